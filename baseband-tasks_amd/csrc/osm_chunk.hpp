// Overlap-save block descriptors handed to the kernels of a plan by value (bbt_kernels.hpp,
// gen_kernels.hpp, gen2_kernels.hpp): which input samples a block reads, which of its samples are
// kept and where they go (reference: PaddedTaskBase._get_frame, baseband_tasks/base.py:775-795).
#pragma once
#if !defined(__HIPCC_RTC__)          // (hipRTC provides the runtime's declarations itself)
#include <hip/hip_runtime.h>
#endif

namespace bbt {

// Blocks dispatched round-robin over the 8 XCDs: give each XCD one contiguous
// range of virtual block ids so neighbours (which share input lines) share an
// L2.  Bijective for any nblocks.
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nblocks) {
    const unsigned q = nblocks >> 3, r = nblocks & 7u;
    const unsigned xcd = bid & 7u, local = bid >> 3;
    const unsigned base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + local;
}

// ---------------------------------------------------------------------------
// Overlap-save block descriptors (one launch handles <= BBT_MAX_CHUNK blocks).
#define BBT_MAX_CHUNK 16
struct OsmBlock {
    long long in_off;   // first input complete sample of the block
    long long out_off;  // output complete sample that receives n == valid_start
    int valid_start;    // first block sample kept
    int valid_count;    // number of block samples kept
    int shift;          // fused channelizer: circular shift o (see k_osm_rowpass), else 0
    int index;          // fused channelizer: index of the block within the call (seam slots)
    int flat;           // k_osm_small only: 1 = the kept range is given in ELEMENTS of the (row, stream)
                        // matrix, not in rows: out_off = element offset in `out` of the first kept
                        // element, valid_count = kept elements, flat_sub = elements of row valid_start
                        // that come before the first kept one (InversePolyphaseFilterBank keeps from
                        // the middle of a row of the block axis)
    int flat_sub;
};

struct OsmChunk {
    int nblk;
    // Regular runs: reg_count > 0 blocks that all look like b[0] with input and output offsets
    // advancing by reg_hop samples per block -- one launch takes any number of them (the
    // descriptor array holds 16); kernels read descriptors through osm_block().
    int reg_count;
    int reg_mask;        // fused channelizer: n_chan - 1 (the blocks' circular shifts step with the output offsets), else 0
    long long reg_hop;
    // Pair-planar hand-over between two plans (bbt_osm_plan_set_layout): a stream of S = 2 P
    // streams stored as P arrays of two-stream samples, pair p at [p * plane, (p + 1) * plane)
    // complete two-stream samples.  out_plane: how the one-kernel plan (k_osm_small) writes its
    // result; in_plane: how the first 256-point column pass reads its input -- 256-byte runs of
    // one pair, as with two streams, instead of 16 bytes out of every 8 S-byte row.  0: interleaved.
    long long in_plane;
    long long out_plane;
    OsmBlock b[BBT_MAX_CHUNK];
};
__device__ __forceinline__ OsmBlock osm_block(const OsmChunk& ch, int i) {
    if (!ch.reg_count) return ch.b[i];
    OsmBlock blk = ch.b[0];
    blk.in_off += i * ch.reg_hop;
    blk.out_off += i * ch.reg_hop;
    if (ch.reg_mask) {
        // shift = (valid_start - out_off) mod n_chan (osm_channelized); seam slots go by the block's index
        blk.shift = (int)(((long long)blk.shift - i * ch.reg_hop) & ch.reg_mask);
        blk.index += i;
    }
    return blk;
}

}  // namespace bbt
