// Host side of fft_gen2.hpp: stage lists, threads per transform and exchange pitches of the
// register-resident mixed-radix transform (modelled in tools/fft_gen2_model.py), its stage
// tables, and the -D options that specialise a kernel on them.  Plain C++ (no device code).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <map>
#include <set>
#include <string>
#include <vector>

namespace bbt {

#define BBT_G2_HOST_MAXS 8
#ifndef BBT_G2_PMAX
#define BBT_G2_PMAX 20                 // points a thread holds at most
#endif
#ifndef BBT_G2_MAXB
#define BBT_G2_MAXB 4                  // butterflies per thread and stage at most
#endif

struct G2Plan {
    int n = 0, nfac = 0;
    int tj = 0;                        // threads per transform
    int ct = 1;                        // interleaved transforms per workgroup (a power of two)
    int fac[BBT_G2_HOST_MAXS] = {};
    int pitch[BBT_G2_HOST_MAXS] = {};  // of the buffer written by stage s (rows of ns[s] fac[s])
    int woff[BBT_G2_HOST_MAXS] = {};   // stage tables: W_{ns R}^{r k} at woff + (r - 1) ns + k
    int table_len = 1;
    int lds_elems = 0;                 // exchange area of the workgroup, 8-byte elements
    int slots = 0;                     // register slots (points) a thread needs
    int pmax = 0;                      // the bound it was planned with
    int threads() const { return std::max(64, (tj * ct + 63) / 64 * 64); }
};

// Points a thread holds at most, per kind of kernel (measured on MI355X, round 5).  The row pass
// of a convolution, eight stages between its load and its store, wants the fewest stages and
// exchanges (20 points: 19.0 against 25.1 us at 10).  The column passes alone run faster with few
// registers and many waves (first column pass of the 1.67 M-point block 16.6 -> 12.8 us at 12),
// but their bound also decides which split N1 x N2 is taken, and over the whole two-lane
// pipeline the six pairs (COL, ROW) in {10, 12, 16, 20} x {16, 20} all measured 34.1-35.5 G (800
// MHz block) and 30.4-31.8 G (600 MHz): within the run-to-run spread, so both stay at 20.  The
// streaming channelizer's optimum depends on the length: its plans time 10, 16 and 20 and keep
// the fastest (bbt_hip.hip, chan_pick); CHAN is what is used untimed.  Environment variables of
// the same names override (dev).
enum { BBT_G2_KIND_CHAN = 0, BBT_G2_KIND_COL = 1, BBT_G2_KIND_ROW = 2 };
static inline int g2_pmax(int kind = BBT_G2_KIND_ROW) {
    static const int v[3] = {
        getenv("BBT_G2_PMAX_CHAN") ? std::max(2, atoi(getenv("BBT_G2_PMAX_CHAN"))) : 10,
        getenv("BBT_G2_PMAX_COL") ? std::max(2, atoi(getenv("BBT_G2_PMAX_COL"))) : BBT_G2_PMAX,
        getenv("BBT_G2_PMAX_ROW") ? std::max(2, atoi(getenv("BBT_G2_PMAX_ROW"))) : BBT_G2_PMAX};
    return v[kind];
}
static inline int g2_maxb(int r, int pmax) { return std::min(BBT_G2_MAXB, std::max(1, pmax / r)); }

// threads one transform needs: every stage's n / R butterflies in rounds of at most MAXB(R)
static inline int g2_threads(int n, const std::vector<int>& fac, int pmax) {
    int t = 1;
    for (int r : fac) t = std::max(t, (n / r + g2_maxb(r, pmax) - 1) / g2_maxb(r, pmax));
    return t;
}

// Radices (2 .. 10, 12, 14, 15, 16; product n): as few stages as possible -- every stage is an
// exchange through LDS with four barriers -- and among those the list needing the fewest threads
// (most points per thread), large radices first (the first stage has no twiddles, and an odd or
// large first radix keeps the padding of the first buffer small).
static inline bool g2_factor(int n, std::vector<int>* out, int pmax) {
    static const int radices[] = {16, 15, 14, 12, 10, 9, 8, 7, 6, 5, 4, 3, 2};
    out->clear();
    if (n < 1) return false;
    if (n == 1) return true;
    struct Best {
        int stages;
        int threads;
        std::vector<int> fac;
    };
    std::map<int, Best> best;
    best[1] = Best{0, 1, {}};
    std::vector<int> divisors;
    for (int d = 1; d <= n; ++d)
        if (n % d == 0) divisors.push_back(d);
    for (int d : divisors) {
        auto it = best.find(d);
        if (it == best.end()) continue;
        const Best cur = it->second;
        for (int r : radices) {
            const int64_t e = (int64_t)d * r;
            if (r > pmax || e > n || n % e) continue;
            Best cand{cur.stages + 1, 0, cur.fac};
            cand.fac.push_back(r);
            std::sort(cand.fac.begin(), cand.fac.end(), std::greater<int>());
            cand.threads = g2_threads(n, cand.fac, pmax);
            auto jt = best.find((int)e);
            if (jt == best.end() || std::make_pair(cand.stages, cand.threads) <
                                        std::make_pair(jt->second.stages, jt->second.threads))
                best[(int)e] = cand;
        }
    }
    auto it = best.find(n);
    if (it == best.end() || it->second.stages > BBT_G2_HOST_MAXS) return false;
    *out = it->second.fac;
    return true;
}

// Passes the LDS needs for one 8-byte access of a wave, by the rules the power-of-two core was
// laid out with (tools/fft_model.py; SQ_LDS_BANK_CONFLICT measured 0 there): ds_read_b64 goes in
// two groups of 32 lanes over 32 eight-byte banks, ds_write_b64 in four groups of 16 lanes over 16.
// addr: element address per lane, < 0 = lane idle.  Returns passes beyond the minimum.
static inline int g2_extra_passes(const int (&addr)[64], bool write) {
    const int group = write ? 16 : 32;
    int extra = 0;
    for (int g0 = 0; g0 < 64; g0 += group) {
        int count[32] = {};
        int seen[32][4];
        int worst = 0;
        for (int l = g0; l < g0 + group; ++l) {
            if (addr[l] < 0) continue;
            const int bank = addr[l] % group;
            bool dup = false;                              // (the same address twice is a broadcast)
            for (int i = 0; i < count[bank] && i < 4; ++i) dup = dup || seen[bank][i] == addr[l];
            if (dup) continue;
            if (count[bank] < 4) seen[bank][count[bank]] = addr[l];
            worst = std::max(worst, ++count[bank]);
        }
        extra += std::max(0, worst - 1);
    }
    return extra;
}

// Geometry for the stage list `fac` (in execution order) with `ct` interleaved transforms per
// workgroup and `tj` threads per transform (0: the fewest the stages allow).
static inline G2Plan g2_make_plan(int n, const std::vector<int>& fac, int ct, int pmax, int tj = 0) {
    G2Plan g;
    g.n = n;
    g.nfac = (int)fac.size();
    g.ct = ct;
    g.tj = tj > 0 ? tj : g2_threads(n, fac, pmax);
    g.pmax = pmax;
    int ns = 1, total = 0, lds = 0;
    for (int s = 0; s < g.nfac; ++s) {
        const int r = fac[s];
        g.fac[s] = r;
        g.woff[s] = total;
        if (s > 0) total += (r - 1) * ns;
        g.slots = std::max(g.slots, ((n / r + g.tj - 1) / g.tj) * r);
        // pitch of the buffer this stage writes: rows of ns * r.  A pad can be searched for (up to 16
        // elements a row: the one that costs the writes of this stage and the reads of the next
        // the fewest extra passes, counted over the first waves of the workgroup), but measured on
        // MI355X (round 5, tools/gen2_bench.hip) padding buys nothing -- rows of 3402 / 8100 points
        // 16.30 / 56.2 us per block padded, 16.37 / 55.6 without -- and costs occupancy when the
        // exchange area crosses a third of the CU's LDS (Channelize(6561): 89 padded, 107
        // Gsamples/s without): the default is no padding.
        const int row = ns * r, rows = n / row;
        int pitch = row;
        if (s + 1 < g.nfac && rows > 1) {
            const int rn = fac[s + 1], mn = n / rn, nsn = row;       // the reading stage
            long best_cost = -1;
            const int waves = std::min(4, (g.tj * ct + 63) / 64);
            // (at most an eighth more LDS than the points need)
            const int pad_max = getenv("BBT_G2_PADMAX") ? atoi(getenv("BBT_G2_PADMAX")) : 0;       // (dev; see below)
            for (int pad = 0; pad <= pad_max && pad * 8 <= std::max(row, 8); ++pad) {
                const int p = row + pad;
                long cost = 0;
                for (int w = 0; w < waves; ++w) {
                    int addr[64];
                    for (int b = 0; b * g.tj < n / r; ++b)
                        for (int e = 0; e < r; ++e) {
                            for (int l = 0; l < 64; ++l) {
                                const int tid = w * 64 + l, col = tid % ct, j = tid / ct + b * g.tj;
                                addr[l] = (tid / ct < g.tj && j < n / r) ? ((j / ns) * p + j % ns + e * ns) * ct + col : -1;
                            }
                            cost += g2_extra_passes(addr, true);
                        }
                    for (int b = 0; b * g.tj < mn; ++b)
                        for (int e = 0; e < rn; ++e) {
                            for (int l = 0; l < 64; ++l) {
                                const int tid = w * 64 + l, col = tid % ct, j = tid / ct + b * g.tj;
                                addr[l] = (tid / ct < g.tj && j < mn)
                                              ? ((j / nsn) * p + j % nsn + e * (mn / nsn) * p) * ct + col : -1;
                            }
                            cost += g2_extra_passes(addr, false);
                        }
                }
                // (a pad is worth LDS only if it saves passes: ties go to the smaller one)
                if (best_cost < 0 || cost < best_cost) {
                    best_cost = cost;
                    pitch = p;
                }
            }
        }
        g.pitch[s] = pitch;
        if (s + 1 < g.nfac) lds = std::max(lds, rows * pitch);
        ns *= r;
    }
    g.table_len = std::max(total, 1);
    g.lds_elems = std::max(lds * ct, 1);
    return g;
}
static inline bool g2_plan(int n, int ct, G2Plan* g, int pmax) {
    std::vector<int> fac;
    if (!g2_factor(n, &fac, pmax)) return false;
    *g = g2_make_plan(n, fac, ct, pmax);
    return true;
}
// the same stages in reversed order (the inverse of a convolution)
static inline G2Plan g2_reversed(const G2Plan& g) {
    std::vector<int> fac(g.fac, g.fac + g.nfac);
    std::reverse(fac.begin(), fac.end());
    return g2_make_plan(g.n, fac, g.ct, g.pmax, g.tj);
}

// N = N1 x N2 for a two-level plan: column transforms of N1 points over tiles of `ct` columns,
// row transforms of N2 points.  Measured on MI355X over every split of the 1 666 980-sample block
// (default arguments at 800 MHz; tools/tune_split.py, round 5): 27 ... 36 Gsamples/s, and what
// separates the splits is how the two kernels' workgroups pack into a CU beside each other -- the
// column and the row pass of the two lanes run at the same time, 16 waves per CU at 128 registers:
// 486 x 3430 and 540 x 3087 (both kernels 4 waves) 35.7 and 34.7, 882 x 1890 (8 and 3) 34.7,
// 490 x 3402 (5 and 4: three column workgroups leave room for no row workgroup) 29.5,
// 980 x 1701 (9 and 3) 27.9 -- then the number of stages, then idle lanes.  So: workgroups of
// 1, 2, 4 or 8 waves first, fewest stages next, fullest waves last.
// Rows per workgroup of the row pass: a short row (N2 of a few hundred points: blocks of a few
// 10^4 samples) needs a dozen threads, and a workgroup of one such row is a wave with three
// quarters of its lanes idle -- 31 104 = 128 x 243 (default arguments at 1400 MHz, DM 10): row pass
// 206 us per launch against 57 and 41 us for the column passes.  So a workgroup takes `ct`
// neighbouring rows k1, interleaved like the columns of a column pass (lanes over the rows first:
// 8 rows x 8 consecutive points = 128-byte runs), as many as fill one wave.  BBT_G2_ROW_CT caps it
// (1: one row per workgroup, as before).
static inline int g2_row_ct(int n2, int pmax) {
    static const int cap = getenv("BBT_G2_ROW_CT") ? std::max(1, atoi(getenv("BBT_G2_ROW_CT"))) : 8;
    std::vector<int> fac;
    if (!g2_factor(n2, &fac, pmax)) return 1;
    const int tj = g2_threads(n2, fac, pmax);
    int ct = 1;
    while (2 * ct <= cap && tj * 2 * ct <= 64) ct *= 2;
    return ct;
}
// Columns per tile of a column pass of a SHORT block (n <= 2^17): a short column (N1 of a few
// dozen points) needs two or four threads, and eight of them are a quarter of a wave reading
// 128-byte runs; 16 or 32 columns fill the wave and read 256- / 512-byte runs.  Measured on the
// 31 104-sample block (MI355X, round 5, BBT_GEN_CT): 36 x 864 with 8 / 16 / 32 columns 31.6 / 38.9 /
// 41.9 G, 64 x 486 37.4 / 40.4 / 40.9, 162 x 192 (14 threads per column) 37.1 / 37.4 / 36.2.
static inline int g2_col_ct(int n1, int pmax, int base = 8) {
    std::vector<int> fac;
    if (!g2_factor(n1, &fac, pmax)) return base;
    const int tj = g2_threads(n1, fac, pmax);
    int ct = base;
    while (ct < 32 && tj * ct < 64) ct *= 2;
    return ct;
}
static inline bool g2_choose_split(int64_t n, int ct, int max_n1, int max_n2, int* n1, int* n2) {
    // (workgroups of up to 4 waves pack a CU whatever their number: for short blocks, whose
    // workgroups are that small, three waves are as good as two or four)
    const bool small = n <= (1 << 17);
    auto pow2_waves = [small](int threads) {
        const int w = threads / 64;
        return w == 1 || w == 2 || w == 4 || w == 8 || (small && w == 3);
    };
    bool found = false;
    double best[5] = {0, 0, 0, 0, 0};
    for (int64_t d = 2; d <= max_n1 && d * 2 <= n; ++d) {
        if (n % d || n / d > max_n2 || n / d < d / 4) continue;
        if (small && d < 16 && n / 16 <= max_n2) continue;      // (as the 16 x N2 plans of power-of-two blocks: no shorter columns)
        G2Plan c, r;
        const int rct = g2_row_ct((int)(n / d), g2_pmax(BBT_G2_KIND_ROW));
        // (short blocks: the widest column tiles first -- the longest runs --, see g2_col_ct)
        const int cct = small ? g2_col_ct((int)d, g2_pmax(BBT_G2_KIND_COL), ct) : ct;
        if (!g2_plan((int)d, cct, &c, g2_pmax(BBT_G2_KIND_COL)) || !g2_plan((int)(n / d), rct, &r, g2_pmax(BBT_G2_KIND_ROW))) continue;
        if (c.lds_elems * 8 > 64 * 1024 || c.threads() > 1024 || r.threads() > 1024) continue;
        const double eff = (double)(c.tj * cct) / c.threads() * (double)(r.tj * rct) / r.threads();
        const double key[5] = {(double)(!pow2_waves(c.threads()) + !pow2_waves(r.threads())),
                               (double)(c.nfac + r.nfac), -(double)cct, -eff, (double)d};
        bool better = !found;
        for (int i = 0; i < 5 && !better; ++i) {
            if (key[i] < best[i]) better = true;
            else if (key[i] > best[i]) break;
        }
        if (better) {
            found = true;
            for (int i = 0; i < 5; ++i) best[i] = key[i];
            *n1 = (int)d;
            *n2 = (int)(n / d);
        }
    }
    return found;
}

// exp(-2 pi i m / n) in double, reduced exactly first: (re, im) as floats
static inline void g2_root(long long m, long long n, float* re, float* im) {
    m %= n;
    if (m < 0) m += n;
    const double a = -2.0 * 3.14159265358979323846264338327950288 * (double)m / (double)n;
    *re = (float)std::cos(a);
    *im = (float)std::sin(a);
    if (m == 0) { *re = 1.f; *im = 0.f; }
    if (4 * m == n) { *re = 0.f; *im = -1.f; }
    if (2 * m == n) { *re = -1.f; *im = 0.f; }
    if (4 * m == 3 * n) { *re = 0.f; *im = 1.f; }
}

// stage tables in the layout of G2Plan::woff: (re, im) pairs
static inline std::vector<float> g2_tables(const G2Plan& g) {
    std::vector<float> h((size_t)g.table_len * 2, 0.f);
    int ns = 1;
    for (int s = 0; s < g.nfac; ++s) {
        if (s > 0)
            for (int r = 1; r < g.fac[s]; ++r)
                for (int k = 0; k < ns; ++k) {
                    const size_t i = (size_t)g.woff[s] + (size_t)(r - 1) * ns + k;
                    g2_root((long long)r * k, (long long)ns * g.fac[s], &h[2 * i], &h[2 * i + 1]);
                }
        ns *= g.fac[s];
    }
    return h;
}

// "BBT_G2_TRAIT(NAME, n, nfac, tj, ct, (fac...), (pitch...))": the geometry as source text
static inline std::string g2_trait_source(const char* name, const G2Plan& g) {
    std::string fac, pitch;
    for (int s = 0; s < g.nfac; ++s) {
        fac += (s ? "," : "") + std::to_string(g.fac[s]);
        pitch += (s ? "," : "") + std::to_string(g.pitch[s]);
    }
    return std::string("BBT_G2_TRAIT(") + name + ", " + std::to_string(g.n) + ", " + std::to_string(g.nfac) + ", " +
           std::to_string(g.tj) + ", " + std::to_string(g.ct) + ", (" + fac + "), (" + pitch + "))\n";
}

}  // namespace bbt
