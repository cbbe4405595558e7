// Prints what the host planner of the generic-length engine (csrc/gen2_host.hpp) decides, as
// JSON lines, for tests/test_gen2_planner.py:   gen2_plan_dump plan <pmax> <n> <ct> ... | split <N> ...
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "gen2_host.hpp"
using namespace bbt;

static void dump(const G2Plan& g) {
    printf("{\"n\": %d, \"tj\": %d, \"ct\": %d, \"threads\": %d, \"slots\": %d, \"lds_elems\": %d, \"table_len\": %d, \"fac\": [",
           g.n, g.tj, g.ct, g.threads(), g.slots, g.lds_elems, g.table_len);
    for (int s = 0; s < g.nfac; ++s) printf("%s%d", s ? ", " : "", g.fac[s]);
    printf("], \"pitch\": [");
    for (int s = 0; s < g.nfac; ++s) printf("%s%d", s ? ", " : "", g.pitch[s]);
    printf("], \"woff\": [");
    for (int s = 0; s < g.nfac; ++s) printf("%s%d", s ? ", " : "", g.woff[s]);
    printf("]}");
}

int main(int argc, char** argv) {
    if (argc < 3) return 2;
    if (!strcmp(argv[1], "plan")) {
        const int pmax = atoi(argv[2]);
        for (int i = 3; i + 1 < argc; i += 2) {
            G2Plan g;
            if (!g2_plan(atoi(argv[i]), atoi(argv[i + 1]), &g, pmax)) { printf("null\n"); continue; }
            printf("{\"forward\": ");
            dump(g);
            printf(", \"reversed\": ");
            dump(g2_reversed(g));
            printf(", \"source\": \"%s\"}\n", "ok");
        }
    } else if (!strcmp(argv[1], "split")) {
        for (int i = 2; i < argc; ++i) {
            int n1 = 0, n2 = 0;
            if (!g2_choose_split(atoll(argv[i]), 8, 1024, 8192, &n1, &n2)) { printf("null\n"); continue; }
            G2Plan c, r;
            // (as bbt_osm_plan_create plans them: short blocks with wider column tiles, short rows several to a workgroup)
            const bool small = atoll(argv[i]) <= (1 << 17);
            g2_plan(n1, small ? g2_col_ct(n1, g2_pmax(BBT_G2_KIND_COL)) : 8, &c, g2_pmax(BBT_G2_KIND_COL));
            g2_plan(n2, g2_row_ct(n2, g2_pmax(BBT_G2_KIND_ROW)), &r, g2_pmax(BBT_G2_KIND_ROW));
            printf("{\"n1\": %d, \"n2\": %d, \"col\": ", n1, n2);
            dump(c);
            printf(", \"row\": ");
            dump(r);
            printf("}\n");
        }
    } else if (!strcmp(argv[1], "source")) {
        G2Plan g;
        if (!g2_plan(atoi(argv[2]), 1, &g, g2_pmax(BBT_G2_KIND_ROW))) return 1;
        const G2Plan gr = g2_reversed(g);
        printf("#include \"gen2_kernels.hpp\"\n%s%sBBT_G2_KERNEL_OSM_SMALL(k_small, GA, GB, 0)\nBBT_G2_KERNEL_FFT_ROWS(k_rows, GA, -1, 0)\n",
               g2_trait_source("GA", g).c_str(), g2_trait_source("GB", gr).c_str());
    } else if (!strcmp(argv[1], "source2")) {
        // the translation unit bbt_osm_plan_create writes for a two-level block (row kernel + both column passes)
        int n1 = 0, n2 = 0;
        const long long n = atoll(argv[2]);
        if (!g2_choose_split(n, 8, 1024, 8192, &n1, &n2)) return 1;
        G2Plan q1, q2;
        const int ct = n <= (1 << 17) ? g2_col_ct(n1, g2_pmax(BBT_G2_KIND_COL)) : 8;
        if (!g2_plan(n2, g2_row_ct(n2, g2_pmax(BBT_G2_KIND_ROW)), &q2, g2_pmax(BBT_G2_KIND_ROW)) ||
            !g2_plan(n1, ct, &q1, g2_pmax(BBT_G2_KIND_COL)))
            return 1;
        const G2Plan q2r = g2_reversed(q2);
        const char* w2 = q2.threads() >= 448 ? "4" : "0";
        const char* w1 = q1.threads() >= 448 ? "4" : "0";
        printf("#include \"gen2_kernels.hpp\"\n%s%s%sBBT_G2_KERNEL_ROW(k_row, GA, GB, %s)\n"
               "BBT_G2_KERNEL_COL(k_first, GC, true, %s)\nBBT_G2_KERNEL_COL(k_last, GC, false, %s)\n",
               g2_trait_source("GA", q2).c_str(), g2_trait_source("GB", q2r).c_str(), g2_trait_source("GC", q1).c_str(),
               w2, w1, w1);
    }
    return 0;
}
