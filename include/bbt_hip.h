/* bbt_hip.h -- C ABI of libbbt_hip.so: MI355X (gfx950) kernels for the
 * coherent-dedispersion -> channelizer hot path of baseband-tasks.
 *
 * The reference (mhvk/baseband-tasks) is pure Python; the seam this library
 * sits behind is the task hook `TaskBase.task(self, data) -> ndarray`
 * (reference baseband_tasks/base.py:699-706) of these operators:
 *
 *   bbt_osm_*   Disperse.task / Dedisperse   dispersion.py:135-139
 *               Convolve.task (and Resample) convolution.py:116-120
 *               i.e.  ifft(fft(x, axis=0) * H, axis=0)[valid]  per
 *               overlap-save block (PaddedTaskBase, base.py:743-795)
 *   bbt_chan_*  Channelize.task / Dechannelize.task
 *               channelize.py:73-74, 164-165  (fft over groups of n samples)
 *   bbt_pfb_*   PolyphaseFilterBankSamples.ppf + Channelize.task
 *               pfb.py:91-100 (definition), 145-154 (Fourier form)
 *
 * Conventions
 *   - every function returns 0 on success, non-zero on failure;
 *     bbt_last_error() then describes the failure (thread-local string).
 *   - no exceptions cross the boundary; handles are opaque.
 *   - array layout is numpy complex64, C-contiguous, shape (n, S): time
 *     major, the S = prod(sample_shape) streams innermost, interleaved
 *     (re, im) float32.  S must be even (callers pad odd S, see
 *     bbt_pad_streams) or, where a function says so, 1.
 *   - `*_dev` pointers are device pointers on the current device; the caller
 *     owns them (bbt_malloc, or any other HIP allocation such as a torch
 *     tensor's data_ptr()).  `stream` is a hipStream_t (NULL = default).
 *   - execution is asynchronous on `stream`; use bbt_stream_sync.
 */
#ifndef BBT_HIP_H
#define BBT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* bbt_stream; /* hipStream_t */
typedef void* bbt_event;  /* hipEvent_t  */
typedef struct bbt_osm_plan bbt_osm_plan;
typedef struct bbt_chan_plan bbt_chan_plan;
typedef struct bbt_pfb_plan bbt_pfb_plan;
typedef struct bbt_shift_plan bbt_shift_plan;
typedef struct bbt_fir_plan bbt_fir_plan;
typedef struct bbt_comm bbt_comm;

/* ---- library / device ------------------------------------------------- */
const char* bbt_last_error(void);
int bbt_version(void);
int bbt_device_count(int* count);
/* Lengths that are not powers of two -- the block and channel counts the reference's own FFT
 * engine picks by default, next_fast_len: baseband_tasks/fourier/numpy.py:99-126, block rule
 * base.py:750-758 -- run on kernels specialised on that length when the plan is made (hipRTC,
 * csrc/rtc.hpp; as rocFFT builds its kernels).  mode: 0 off (BBT_RTC=0: the general kernels run
 * every length), 1 on with a fall back to them, 2 required (BBT_RTC=require); modules: code
 * objects compiled by this process so far, seconds: the time that took.  BBT_RTC_CACHE=<dir> keeps
 * them on disk, BBT_CSRC=<dir> says where the kernel headers are if not next to the library. */
int bbt_rtc_info(int* mode, int64_t* modules, double* seconds);
/* Plans whose best kernel depends on the length in no regular way -- Channelize / Dechannelize with
 * channel counts that are not powers of two (channelize.py:73-74 on fourier/numpy.py:99-126) and
 * PolyphaseFilterBank on 8 streams and more (pfb.py:136-154) -- time their candidates once when
 * they are made, on scratch device memory the library keeps per device (2 GiB, at most a quarter
 * of what is free when it is first needed).  bytes: what the current device holds now; release
 * != 0 frees it (the next such plan allocates it again).  BBT_TUNE_KEEP=0 frees it after every plan. */
int bbt_tune_scratch(int release, int64_t* bytes);
int bbt_set_device(int device);
int bbt_get_device(int* device);
int bbt_device_name(char* buf, int buflen);

/* ---- memory, streams, events (plumbing) -------------------------------- */
/* Device memory from a caching pool: bbt_free keeps the block for reuse by a
 * later bbt_malloc of about the same size (no hipFree, hence no device
 * synchronisation between consecutive reader calls).  Takes the place of the
 * np.empty the reference's Base.read does per call (base.py:416).
 * RULE: blocks of the pool are used on ONE stream at a time, the "pool
 * stream" (bbt_pool_set_stream; default NULL = the default stream), from one
 * thread at a time: bbt_free returns a block at once, without an event, so its
 * reuse is ordered only by that stream.  Changing the pool stream drains the
 * device once (a buffer whose owner is garbage collected after the switch may
 * still have had kernels queued on the previous stream).  Work on other streams
 * must be synchronised by the caller before its buffers are freed (plans join
 * their internal streams before an execute call returns).
 * BBT_POOL=0 disables caching, BBT_POOL_MAX_GB caps the idle bytes kept
 * (default 96). */
int bbt_malloc(void** dev_ptr, size_t nbytes);
int bbt_free(void* dev_ptr);
int bbt_pool_set_stream(bbt_stream stream);               /* the stream pool blocks are used on */
int bbt_pool_trim(void);                                   /* hipFree every idle block */
int bbt_pool_info(int64_t* cached_bytes, int64_t* live_bytes);
int bbt_host_alloc(void** host_ptr, size_t nbytes); /* pinned */
int bbt_host_free(void* host_ptr);
/* Page-lock memory the caller owns (a NumPy array, a memory map), so that the
 * copies below are asynchronous DMA transfers from / to it; undo before the
 * memory is released.  Together with bbt_stream_wait_event these carry the
 * host path of Base.read (base.py:389-438): upload of block run m + 1,
 * transforms of run m and download of run m - 1 at the same time. */
int bbt_host_register(void* host_ptr, size_t nbytes);
int bbt_host_unregister(void* host_ptr);
int bbt_memset(void* dev_ptr, int value, size_t nbytes, bbt_stream stream);
int bbt_memcpy_h2d(void* dst_dev, const void* src_host, size_t nbytes, bbt_stream stream);
int bbt_memcpy_d2h(void* dst_host, const void* src_dev, size_t nbytes, bbt_stream stream);
int bbt_memcpy_d2d(void* dst_dev, const void* src_dev, size_t nbytes, bbt_stream stream);
/* 2-D copy (rows of `width` bytes, given pitches); kind: 0 h2d, 1 d2h, 2 d2d.
 * Used to pad an odd stream count to even and to strip the pad again. */
int bbt_memcpy2d(void* dst, size_t dpitch, const void* src, size_t spitch, size_t width,
                 size_t height, int kind, bbt_stream stream);
/* out[r, c] = c < n_in ? in[r, c] : 0 for rows of n_in -> n_out elements of 4 or 8
 * bytes: an odd stream count padded to even in one pass (the kernels take
 * stream pairs). */
int bbt_pad_streams(const void* in_dev, void* out_dev, int64_t n_rows, int n_in, int n_out,
                    int elem_bytes, bbt_stream stream);
int bbt_stream_create(bbt_stream* stream);
int bbt_stream_destroy(bbt_stream stream);
int bbt_stream_sync(bbt_stream stream);
int bbt_device_sync(void);
int bbt_event_create(bbt_event* ev);
/* An event that only orders streams of this device among each other (bbt_stream_wait_event):
 * no timing, and no system-scope fence when it is recorded -- a default event writes the caches
 * back and invalidates them, which empties the Infinity Cache under whatever runs next.  Not for
 * bbt_event_sync / bbt_event_elapsed_ms by a host that then reads device results. */
int bbt_event_create_ordering(bbt_event* ev);
int bbt_event_destroy(bbt_event ev);
int bbt_event_record(bbt_event ev, bbt_stream stream);
int bbt_event_sync(bbt_event ev);
/* *done = 1 if everything in front of the last record of `ev` has finished, else 0; never blocks.
 * (The Python layer lets go of the plans and buffers that finished deferred calls still hold with
 * it -- a deferred call's completion event, bbt_osm_plan_defer -- instead of keeping them until
 * somebody waits: hip.py `_prune`.) */
int bbt_event_query(bbt_event ev, int* done);
int bbt_stream_wait_event(bbt_stream stream, bbt_event ev);   /* later work on `stream` waits for `ev` */
int bbt_event_elapsed_ms(bbt_event start, bbt_event stop, float* ms);

/* ---- overlap-save spectral multiply: Dedisperse / Disperse / Convolve ---
 * Replaces Disperse.task (dispersion.py:135-139) and Convolve.task
 * (convolution.py:116-120).
 *
 *   n_fft       block length N (= PaddedTaskBase._ih_samples_per_frame),
 *               a power of two, 256 <= N <= 2^24 (one kernel up to 4096 and
 *               for 8192 and 16384, two-level four-step up to 2^22, three-level
 *               above): the fast
 *               path; or any N = 2^a 3^b 5^c 7^d -- the lengths the
 *               reference's NumPy engine picks (fourier/numpy.py:99-126,
 *               block rule base.py:750-758) -- that is <= 8192 or splits
 *               into two such factors <= 8192 (generic LDS Stockham path)
 *   n_stream    S, even; or 1 with a power-of-two n_fft: the one stream runs
 *               unpadded, two consecutive blocks side by side where a pair of
 *               streams would be (bbt_osm_execute, and
 *               bbt_osm_execute_channelized for 256 channels and up; no fused
 *               detection)
 *   n_resp      number of distinct response columns C
 *   resp        C x N complex64, FFT-natural order, UNSCALED
 *               (= Disperse.phase_factor, dispersion.py:115-129, or
 *               Convolve._ft_response, convolution.py:108-114); the 1/N of
 *               the inverse transform is applied by the library
 *   resp_on_device  0: `resp` is a host pointer; 1: device pointer
 *   resp_index  S ints: response column used by each stream (NULL = all 0)
 * A plan runs one execute call at a time: its work buffers, seam buffer and
 * events belong to the running call, so calls on one plan are serialised by a
 * mutex inside it on the host and, on the device, a call first makes its stream
 * wait for the end of the previous call (which may have run on another stream;
 * consecutive deferred calls, bbt_osm_plan_defer, are the one exception).
 * Calls on different plans are independent.
 * bbt_osm_plan_create itself synchronises the device when
 * `resp` is a device pointer, so a response still being written on another
 * stream is complete before it is permuted.
 */
int bbt_osm_plan_create(bbt_osm_plan** plan, int64_t n_fft, int n_stream, int n_resp,
                        const void* resp, int resp_on_device, const int32_t* resp_index);
int bbt_osm_plan_destroy(bbt_osm_plan* plan);
/* Bytes of device workspace the plan holds, and the number of blocks it
 * processes per kernel launch. */
int bbt_osm_plan_info(const bbt_osm_plan* plan, int64_t* workspace_bytes, int* chunk_blocks,
                      int* n1, int* n2);
/* Deferred join for the NEXT bbt_osm_execute* call on `plan` (that one call only, whatever
 * becomes of it).  By default an execute call returns with `stream` ordered after all of its
 * work: anything queued on `stream` afterwards sees the results.  A plan with lanes (more than
 * one kernel per block: bbt_osm_plan_info reports n1 > 1) runs its kernels on internal streams, so that default costs a drain at
 * every call boundary -- the lanes run empty, `stream` takes over, the next call's lanes start
 * again: 2-4 % of a 768-block call on MI355X.  A reader that takes consecutive frames
 * (base.py:427-436) does not need `stream` ordered after call i to issue call i + 1.  With a
 * completion event handed in, the call leaves `stream` as it found it and records `done` where
 * its work really ends (an internal stream, after the lanes and the seam pass of the fused
 * channelizer): whoever consumes `out_dev` -- or frees or overwrites `in_dev` or `out_dev` --
 * must first wait for `done` (bbt_stream_wait_event, bbt_event_sync).  Consecutive deferred
 * calls on one plan flow into each other: lane order protects the work buffers, the seam
 * buffer has two turns.  A plan without lanes (one kernel per chunk: n_fft <= 4096, 8192 and 16384,
 * or <= 8192 on the generic path) runs a deferred call on an internal stream of its own, after what was
 * queued on `stream` before it and beside what is queued there next -- the upstream task's
 * kernels for the following run (InversePolyphaseFilterBank: Dechannelize of run k + 1 beside the
 * deconvolution of run k).  The isolated timing mode simply records `done` on `stream`.
 * done == NULL cancels.
 * The Python host layer uses this for every plan call whose output it owns (hip.py: the event
 * travels with the DeviceArray and is waited for by the next thing that touches it). */
int bbt_osm_plan_defer(bbt_osm_plan* plan, bbt_event done);
/* 1 if bbt_osm_execute_channelized can take Channelize(n_chan) into this
 * plan's row pass (power-of-two block of two or three levels, n_chan a power
 * of two in [256, 4096] dividing the row length), else 0. */
int bbt_osm_plan_fusable(const bbt_osm_plan* plan, int n_chan);
/* Process n_blocks overlap-save blocks.  Block b reads input complete
 * samples [in_off[b], in_off[b] + N) of `in_dev` and writes block samples
 * [valid_start[b], valid_start[b] + valid_count[b]) to output complete
 * samples starting at out_off[b] of `out_dev` (all offsets in complete
 * samples; arrays are host arrays of length n_blocks).  This is
 * PaddedTaskBase._seek_frame/_read_frame + task()[pad_slice]
 * (base.py:775-795, dispersion.py:139) for a batch of frames. */
int bbt_osm_execute(bbt_osm_plan* plan, const void* in_dev, void* out_dev, int64_t n_blocks,
                    const int64_t* in_off, const int64_t* out_off, const int32_t* valid_start,
                    const int32_t* valid_count, bbt_stream stream);
/* bbt_osm_execute with the kept range given in ELEMENTS of the (row, stream)
 * matrix instead of whole rows, for plans of one kernel (power-of-two n_fft <=
 * 4096, 8192 or 16384; S even): block b keeps valid_elems[b] elements starting at element
 * first_elem (even, < S) of row valid_start[b], written contiguously to out_dev
 * from element out_elem_off[b].  InversePolyphaseFilterBank (pfb.py:255-269) runs
 * its transform along the block axis with one stream per polyphase phase and
 * keeps from the middle of a row: ifft(...)[pad_slice] on the flattened frame. */
int bbt_osm_execute_flat(bbt_osm_plan* plan, const void* in_dev, void* out_dev, int64_t n_blocks,
                         const int64_t* in_off, const int64_t* out_elem_off, const int32_t* valid_start,
                         int32_t first_elem, const int32_t* valid_elems, bbt_stream stream);
/* Pair-planar hand-over between two plans of the same S = 2 P streams (a short filter in
 * front of a dispersion, `Dedisperse(Resample(x))`: sampling.py:308-312 + dispersion.py:135-139):
 * the intermediate stream, which only the two plans see, is laid out as P arrays of two-stream
 * samples -- pair p at complete two-stream samples [p * plane, (p + 1) * plane) -- so that the
 * consumer's first column pass reads 256-byte runs of one pair instead of 16 bytes out of every
 * 8 S-byte row.  out_plane > 0 (one-kernel plans: n_fft <= 4096, 8192, 16384): bbt_osm_execute /
 * bbt_osm_execute_regular write sample r of pair p at out_dev[(p * out_plane + r) * 2 ...]
 * (complex64 units; offsets in the block descriptors count samples of a plane).  in_plane > 0
 * (two-level plans with 256-point columns, n_fft 2^17 ... 2^20): the executes read their input
 * that way.  0, 0 restores the interleaved (n, S) layout.  The setting holds until changed. */
int bbt_osm_plan_set_layout(bbt_osm_plan* plan, int64_t in_plane, int64_t out_plane);
/* Fused Channelize(Dedisperse(...), n_chan): as bbt_osm_execute, but instead
 * of the dedispersed samples it writes their channelization
 * (Channelize.task, channelize.py:73-74): spectrum s is the unnormalised FFT
 * of output complete samples [s*n_chan, (s+1)*n_chan) of the stream the blocks
 * would have produced.  Spectra first_spectrum .. first_spectrum+n_spectra-1
 * are stored to out_dev as (n_spectra, n_chan, S); blocks must be consecutive
 * in the output stream (out_off[b+1] == out_off[b] + valid_count[b]) wherever a
 * wanted spectrum straddles them.  n_chan: whatever bbt_osm_plan_fusable
 * accepts (a power of two, 256 <= n_chan <= row length n2 of
 * bbt_osm_plan_info, or 16..128 for blocks with 256 columns or of three
 * levels; power-of-two n_fft of more than one kernel: 2^15 and up), valid_count >= n_chan; valid_start may be
 * anything, 0 included.  The dedispersed stream itself never exists in memory. */
int bbt_osm_execute_channelized(bbt_osm_plan* plan, const void* in_dev, void* out_dev,
                                int64_t n_blocks, const int64_t* in_off, const int64_t* out_off,
                                const int32_t* valid_start, const int32_t* valid_count, int n_chan,
                                int64_t first_spectrum, int64_t n_spectra, bbt_stream stream);
/* Fused Integrate(Power|Square(Channelize(Dedisperse(...), n_chan)), step)
 * (functions.py:15-16, 131-143; integration.py:252-303 with an integer step):
 * as bbt_osm_execute_channelized, but the spectra first_spectrum ..
 * first_spectrum + n_bins*step - 1 are detected and summed in the epilogue of
 * the last column pass instead of being stored, so the channelized stream never
 * exists either.  out_dev: float32 (n_bins, n_chan, S/2, 4) for mode 1 (|X|^2,
 * |Y|^2, Re XY*, Im XY* of each stream pair) or (n_bins, n_chan, S) for mode 0
 * (|z|^2 per stream); average != 0 divides by step.  Sums are formed with float
 * atomics, so the last bits depend on the order of arrival.  Needs block lengths
 * 2^16..2^20 and bbt_osm_detect_bins_max(plan, n_chan, step) <= 64 (e.g. step
 * >= 17 for 1024 channels on 2^20-sample blocks).  step == 1 is
 * Power|Square(Channelize(...)) without integration: every power is stored where
 * its spectrum would have gone (plain stores: deterministic, no zeroing of out_dev,
 * no limit on bins). */
int bbt_osm_execute_channelized_detect(bbt_osm_plan* plan, const void* in_dev, void* out_dev,
                                       int64_t n_blocks, const int64_t* in_off,
                                       const int64_t* out_off, const int32_t* valid_start,
                                       const int32_t* valid_count, int n_chan,
                                       int64_t first_spectrum, int64_t n_bins, int step, int mode,
                                       int average, bbt_stream stream);
int bbt_osm_detect_bins_max(const bbt_osm_plan* plan, int n_chan, int step);
/* Regular case: block b has in_off = in_off0 + b*hop, out_off = out_off0 +
 * b*hop, the same valid_start and valid_count = hop. */
int bbt_osm_execute_regular(bbt_osm_plan* plan, const void* in_dev, void* out_dev,
                            int64_t n_blocks, int64_t in_off0, int64_t out_off0, int64_t hop,
                            int32_t valid_start, bbt_stream stream);
/* Per-pass timing with HIP events on the stream each pass is launched on
 * (off by default).  enable: 0 off; 1 time the normal schedule (the lanes
 * stay on, so a pass may share the GPU with a pass of the other lane -- this
 * is what rocprofv3 --kernel-trace sees); 2 isolated: single lane, passes run
 * one after the other.  ms[0..2] = accumulated column-forward / row /
 * column-inverse pass time, launches = launches of each pass since enabling. */
int bbt_osm_timing_enable(bbt_osm_plan* plan, int enable);
int bbt_osm_timing_read(bbt_osm_plan* plan, double ms[3], int64_t* launches);
/* The same per pass, with the number of timed launches and the overlap-save
 * blocks they covered: only every BBT_OSM_TIMING_STRIDE-th (4) chunk of a lane
 * carries events (they cost queue slots of their own), so the per-launch mean
 * is ms[k] / launches[k] and the time per block ms[k] / blocks[k]. */
int bbt_osm_timing_read_passes(bbt_osm_plan* plan, double ms[3], int64_t launches[3], int64_t blocks[3]);

/* ---- channelizer: Channelize / Dechannelize -----------------------------
 * Replaces Channelize.task (channelize.py:73-74): FFT over each group of
 * n_chan consecutive complete samples; in (n_spectra*n_chan, S) ->
 * out (n_spectra, n_chan, S).  direction -1: forward, unnormalised
 * (Channelize); +1: inverse, scaled by 1/n_chan (Dechannelize,
 * channelize.py:164-165).  n_chan a power of two, 2..4096 (fast path), 8192 or
 * 16384 (one workgroup of 512 / 1024 threads per transform: stream pairs, directions
 * -1 / +1), or any 2^a 3^b 5^c 7^d <= 8192.  n_stream even, or 1 for a power-of-two n_chan
 * in [256, 4096] (one stream: two consecutive groups are transformed side by
 * side, nothing is padded).  direction -2: every stream is z = a + i b of two
 * real streams and out receives their half spectra, (n_spectra, n_chan/2 + 1,
 * 2 n_stream) -- Channelize of float32 streams in one pass;
 * direction +2 is its inverse (half spectra in, the streams z out, scaled by 1/n_chan). */
int bbt_chan_plan_create(bbt_chan_plan** plan, int n_chan, int n_stream, int direction);
int bbt_chan_plan_destroy(bbt_chan_plan* plan);
int bbt_chan_execute(bbt_chan_plan* plan, const void* in_dev, void* out_dev, int64_t n_spectra,
                     bbt_stream stream);

/* ---- polyphase filter bank ----------------------------------------------
 * Replaces PolyphaseFilterBank(Samples).ppf + Channelize.task
 * (pfb.py:91-100, 145-154): out[i] = FFT_c( sum_t in[(i+t)*n_chan + c] *
 * taps[t, c] ).  Reads (n_spectra + n_tap - 1) * n_chan input samples.
 * taps: host float32 (n_tap, n_chan).  n_stream even, or 1 for n_chan in
 * 256..2048 with 4, 8, 12 or 16 taps (the sliding-window kernels: two groups of
 * spectra of the one stream side by side); n_stream -S (S = 1 or even): S
 * streams z = a + i b made of two real streams each, out receiving their half
 * spectra (n_spectra, n_chan/2 + 1, 2 S) (sliding-window kernels only). */
int bbt_pfb_plan_create(bbt_pfb_plan** plan, int n_tap, int n_chan, int n_stream,
                        const float* taps_host);
int bbt_pfb_plan_destroy(bbt_pfb_plan* plan);
int bbt_pfb_execute(bbt_pfb_plan* plan, const void* in_dev, void* out_dev, int64_t n_spectra,
                    bbt_stream stream);

/* ---- detection and integration -------------------------------------------
 * Replaces Square.task / Power.task (functions.py:15-16, 131-143) and, for an
 * integer step, Integrate._read_frame (integration.py:252-303), in one pass:
 * out[i] = (average ? 1/step : 1) * sum_{s<step} f(in[i*step + s]).
 *   mode 0  f = |z|^2 per complex element: in (n_out*step, n_elem) complex64,
 *           out (n_out, n_elem) float32
 *   mode 1  f = |X|^2, |Y|^2, Re(X Y*), Im(X Y*) for consecutive (X, Y) element
 *           pairs: out (n_out, n_elem/2, 4) float32
 *   mode 2  f = identity on float32 elements: in (n_out*step, n_elem) float32 */
int bbt_detect_integrate(const void* in_dev, void* out_dev, int64_t n_out, int64_t step,
                         int64_t n_elem, int mode, int average, bbt_stream stream);
/* Mode 1 with the polarization axis anywhere in the sample (Power takes any
 * axis, functions.py:131-143): in (n_out*step, outer, 2, inner) complex64, X =
 * [.., 0, :], Y = [.., 1, :]; out (n_out, outer, 4, inner) float32. */
int bbt_detect_power_axis(const void* in_dev, void* out_dev, int64_t n_out, int64_t step, int outer,
                          int inner, int average, bbt_stream stream);

/* ---- integer sample shifts -------------------------------------------------
 * Replaces ShiftSamples.task (sampling.py:424-425, data[self._indices]), the
 * base of DisperseSamples / DedisperseSamples (dispersion.py:193-298):
 * out[i, e] = in[i + offsets[e], e] for the n_elem elements (4 or 8 bytes each)
 * of a complete sample, offsets[e] = shift.max() - shift[e] >= 0.  `in` must
 * hold n_out + max(offsets) complete samples. */
int bbt_shift_plan_create(bbt_shift_plan** plan, int n_elem, int elem_bytes,
                          const int32_t* offsets_host);
int bbt_shift_plan_destroy(bbt_shift_plan* plan);
int bbt_shift_execute(bbt_shift_plan* plan, const void* in_dev, void* out_dev, int64_t n_out,
                      bbt_stream stream);

/* ---- real-valued streams ------------------------------------------------------
 * float32 streams run through the complex kernels (the reference's rfft/irfft
 * engine paths, fourier/numpy.py:41-49).  Element-wise glue, n_total = number
 * of OUTPUT elements:
 *   op 0  real -> complex (zero imaginary part)
 *   op 1  complex -> real part
 *   op 2  half spectrum (n_total/(n_chan*n_stream) spectra of n_chan/2+1
 *         channels, n_stream streams innermost) -> Hermitian full spectrum of
 *         n_chan channels as irfft interprets it
 *   op 3  x -> x*x
 *   op 4  transform of z = a + i b (n_total/((n_chan/2+1)*n_stream) spectra of
 *         n_chan channels, n_stream/2 complex streams) -> the half spectra of the
 *         n_stream real streams a, b, ... (n_chan/2+1 channels): two real
 *         transforms for the price of one complex one
 *   op 5  its inverse: half spectra of n_stream real streams -> the n_chan-channel
 *         spectra of n_stream/2 complex streams z = a + i b (n_total counts those)
 *   op 6  as op 4 for an input that carries n_stream/2 + 1 complex streams, the
 *         last one unused (an odd number of pairs transformed padded to even) */
int bbt_real_op(const void* in_dev, void* out_dev, int op, int64_t n_total, int n_chan,
                int n_stream, bbt_stream stream);

/* The dispersion chirp on the GPU: Disperse.phase_factor (dispersion.py:115-129) with
 * DispersionMeasure.phase_delay (dm.py:78-105), float64 arithmetic cast to complex64 as
 * the reference casts it.  For column c and FFT bin k of n (numpy.fft.fftfreq order):
 *   f = freq_hz[c] + sideband[c] * fftfreq(n, 1 / rate_hz)[k]
 *   phase = sideband[c] * d_dm * f_MHz * (1 / ref_MHz[c] - 1 / f_MHz)^2 * 1e6 + offset_s * fftfreq[k]   (cycles)
 *   out_dev[c * n + k] = exp(2 pi i phase)
 * d_dm = dispersion constant (1 / 2.41e-4 s MHz^2 cm^3 / pc) times DM, negative to dedisperse;
 * offset_s = sample_offset / rate (reference frequency outside the band, dispersion.py:123-125).
 * out_dev: (n_col, n) complex64, what bbt_osm_plan_create takes as a device-resident response.
 * The host arrays hold n_col doubles each.  Synchronises `stream`. */
int bbt_chirp(void* out_dev, int64_t n, int n_col, const double* freq_hz, const double* sideband,
              const double* ref_hz, double rate_hz, double d_dm, double offset_s, bbt_stream stream);

/* Per-stream complex factor: out[i, e] = in[i, e] * factor_dev[e] for the n_elem
 * complex64 elements of a complete sample (TimeDelay.task, sampling.py:374-377).
 * In-place (out_dev == in_dev) is allowed. */
int bbt_scale_streams(const void* in_dev, void* out_dev, int64_t n_samples, int n_elem,
                      const void* factor_dev, bbt_stream stream);

/* ---- short responses in the time domain ---------------------------------------
 * Convolve.task (convolution.py:116-120) keeps
 * ifft(fft(x) * fft(response))[n_tap-1:], the exact linear convolution
 *   out[i, s] = sum_k response[k, s] * in[i + n_tap - 1 - k, s];
 * for short responses (the 2*pad+1 = 129-tap windowed sinc of ShiftAndResample /
 * Resample, sampling.py:177-193) this computes it directly: one read and one
 * write of the stream instead of three FFT passes, and no block structure.
 * response_host: complex64 (n_tap, n_stream), the reference's `response`
 * broadcast to the streams (n_stream even, or 1: the two halves of the one
 * stream's time range are then filtered side by side).  `in` holds n_out +
 * n_tap - 1 complete samples.  Purely real responses take a cheaper kernel. */
int bbt_fir_plan_create(bbt_fir_plan** plan, int n_tap, int n_stream, const void* response_host);
int bbt_fir_plan_destroy(bbt_fir_plan* plan);
int bbt_fir_execute(bbt_fir_plan* plan, const void* in_dev, void* out_dev, int64_t n_out,
                    bbt_stream stream);

/* ---- sampler frames ----------------------------------
 * Device-side decode of packed VDIF / DADA payloads in front of the path (what
 * the `baseband` readers the reference is fed with do on the host; the
 * reference itself has no decoder and no sample files: parity unpinned).
 * Frame f (frame_bytes each, payload after header_bytes) belongs to thread
 * f % n_thread of frame set f / n_thread and holds samples_per_frame complete
 * samples of n_elem components (channels x 1 or 2) of `bits` bits, first
 * sample in the least significant bits of little-endian 32-bit words.
 * out: float32 [(set * samples_per_frame + t), thread, component] -- complex64
 * when the components are (I, Q) pairs.  code 0: VDIF levels (1, 2, 4 bits;
 * 8, 16 bits offset binary); code 1: two's complement (8, 16 bits; DADA). */
int bbt_unpack(const void* raw_dev, void* out_dev, int64_t n_frames, int frame_bytes,
               int header_bytes, int bits, int samples_per_frame, int n_thread, int n_elem,
               int code, bbt_stream stream);
/* The same with one byte per frame in valid_dev (device; NULL = all valid): frames
 * flagged 0 -- invalid in their header, or missing from the file and put in by
 * the host as padding -- are written as zeros (the fill value of `baseband`'s
 * readers). */
int bbt_unpack_masked(const void* raw_dev, void* out_dev, int64_t n_frames, int frame_bytes,
                      int header_bytes, int bits, int samples_per_frame, int n_thread, int n_elem,
                      int code, const void* valid_dev, bbt_stream stream);

/* ---- multi-GPU: one process per GPU, RCCL over xGMI ----------------------
 * The reference has no distributed code; these are what SURVEY 8(b)/(e) ask a
 * replacement to export for the way this path shards (independent overlap-save
 * blocks, base.py:783-790, or independent sub-bands; results concatenated as
 * combining.Concatenate would, combining.py:176-211).  No data-path
 * collective exists: only the response broadcast at plan time and the optional
 * gather of outputs.  librccl is loaded on first use.
 *   bbt_comm_unique_id  rank 0 makes the 128-byte id and hands it to the other
 *                       ranks by any side channel (file, socket, a torch
 *                       store); then every rank calls bbt_comm_init after
 *                       bbt_set_device
 *   bbt_bcast_chirp     in place broadcast of n_complex complex64 from `root`
 *                       (the C x N response later given to
 *                       bbt_osm_plan_create(..., resp_on_device = 1, ...))
 *   bbt_gather_output   all-gather: rank r's bytes_per_rank bytes land at
 *                       recv_dev + r * bytes_per_rank on every rank
 */
#define BBT_COMM_ID_BYTES 128
int bbt_comm_unique_id(void* id_out, size_t id_bytes);
int bbt_comm_init(bbt_comm** comm, int n_ranks, int rank, const void* id, size_t id_bytes);
int bbt_comm_destroy(bbt_comm* comm);
int bbt_bcast_chirp(bbt_comm* comm, void* resp_dev, int64_t n_complex, int root, bbt_stream stream);
int bbt_gather_output(bbt_comm* comm, const void* send_dev, void* recv_dev, int64_t bytes_per_rank,
                      bbt_stream stream);

#ifdef __cplusplus
}
#endif
#endif /* BBT_HIP_H */
