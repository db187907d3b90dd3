/*
 * rbf.h -- C ABI of librbf_hip.so: the MI355X (gfx950) rational-Bloom-filter residual coder.
 *
 * The reference (ross39/new_bloom_filter_repo) is pure Python and has no FFI layer; its
 * boundary for this path is a Python class surface.  Each entry point below names the
 * reference code whose BODY it replaces (file:line under the reference tree); the Python
 * classes in new_bloom_filter_repo_amd/ keep the reference's names and call these through
 * ctypes (INTEGRATION.md shows the stub a reference maintainer would add).
 *
 * Conventions
 *  - Every function returns 0 (RBF_OK) or a negative errno-style code; it never throws and
 *    never aborts.  rbf_last_error() returns a thread-local, library-owned description of
 *    the most recent failure on the calling thread.
 *  - Pointers named *_dev are DEVICE pointers (HIP global memory); everything else is host
 *    memory owned by the caller.  The library owns what it allocates (context scratch,
 *    rbf_malloc blocks until rbf_free) and nothing it returns outlives the context.
 *  - Bit vectors (masks, filters, witnesses) are packed MSB-first per byte, i.e. exactly
 *    numpy.packbits order: stream bit i lives in byte i>>3 at bit 7-(i&7).  Buffers holding
 *    them must be padded to a multiple of 8 bytes; pad bits are written as 0.
 *  - All work of one context is enqueued on ONE HIP stream (the caller's, or one the library
 *    creates); calls on one context are serialised, different contexts are independent.
 *    One context per thread.  Calls return after ENQUEUEING unless stated otherwise.
 *  - Limits: 1 <= n < 2^32 pixels per frame; 1 <= m <= 2^32-1 filter bits; floor_k <= 64.
 */
#ifndef RBF_H
#define RBF_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RBF_OK        0
#define RBF_EINVAL  (-22)
#define RBF_ENOMEM  (-12)
#define RBF_EIO      (-5)   /* a HIP runtime call failed; see rbf_last_error() */
#define RBF_ERANGE  (-34)

#define RBF_ABI_VERSION 4

typedef struct rbf_ctx rbf_ctx;

/* One filter's geometry, computed on the HOST (float64 via libm must match CPython's `math`
 * to the last ulp -- improved_video_compressor.py:181-196 -- so it never runs on the device).
 *   m          filter length in bits            (l,         improved_video_compressor.py:193)
 *   floor_k    deterministic hash count         (floor(k*), :57)
 *   threshold  activate the extra hash iff XXH64(str(i), act_seed) < threshold
 *              (integer form of `h / (2**64-1) < k* - floor(k*)`, :94-97)                  */
typedef struct {
    uint32_t m;
    uint32_t floor_k;
    uint64_t threshold;
} rbf_filter_params;

/* Hash seeds: (0x12345678, 0x87654321, 999) improved_video_compressor.py:62-63,94;
 * (0, 1, 999) bloom_compress.py:163-164,195; (0, 1, ceil(k*)) rational_bloom_filter.py:100-101,134. */
typedef struct {
    uint64_t h1;
    uint64_t h2;
    uint64_t act;
} rbf_seeds;

/* Per-frame results written by encode (device memory, 4 x uint64 per frame). */
#define RBF_STAT_WITNESS_BITS 0   /* len(witness),            improved_video_compressor.py:253 */
#define RBF_STAT_FILTER_ONES  1   /* popcount of the filter */
#define RBF_STAT_RESERVED0    2
#define RBF_STAT_RESERVED1    3
#define RBF_STATS_PER_FRAME   4

/* ---- library / context ------------------------------------------------------------------ */
int rbf_version(void);
const char *rbf_last_error(void);
int rbf_device_count(int *count);

/* hip_stream: a hipStream_t to enqueue on (e.g. torch.cuda.current_stream().cuda_stream),
 * or NULL to let the library create and own a stream. */
int rbf_ctx_create(int device, void *hip_stream, rbf_ctx **out);
int rbf_ctx_destroy(rbf_ctx *ctx);
int rbf_ctx_sync(rbf_ctx *ctx);                       /* blocks until the stream is idle */

/* Device-memory helpers so host code needs no other GPU runtime. */
int rbf_malloc(rbf_ctx *ctx, size_t bytes, void **out_dev);
int rbf_free(rbf_ctx *ctx, void *ptr_dev);
int rbf_memset(rbf_ctx *ctx, void *dst_dev, int value, size_t bytes);            /* async */
int rbf_memcpy_h2d(rbf_ctx *ctx, void *dst_dev, const void *src, size_t bytes);  /* blocks */
int rbf_memcpy_d2h(rbf_ctx *ctx, void *dst, const void *src_dev, size_t bytes);  /* blocks */
int rbf_memcpy_d2d(rbf_ctx *ctx, void *dst_dev, const void *src_dev, size_t bytes);   /* async, on the context's stream */

/* Per-kernel HIP-event timing (bench.py): launches of the selected kernels are bracketed by
 * events on the context's stream.  `on`: 0 = off, 1 = every kernel, otherwise a bit mask with
 * bit RBF_K_* set for each kernel to time.  rbf_timing_read blocks until the stream is idle. */
#define RBF_K_MASK    0
#define RBF_K_INSERT  1
#define RBF_K_QUERY   2
#define RBF_K_STITCH  3
#define RBF_K_EXPAND  4
#define RBF_K_GATHER  5
#define RBF_K_SCATTER 6
#define RBF_K_INDEX   7
#define RBF_K_REDUCE  8
#define RBF_K_SCAN    9
#define RBF_K_NOISE   10
#define RBF_K_PACK    11
#define RBF_K_HASHTAB 12      /* k_hash_table: the per-batch table of the pixel indices' three hashes */
#define RBF_K_COUNT   13
int rbf_timing_enable(rbf_ctx *ctx, int on);
/* Testing / tuning knob (bit mask).  0 (default) = pick the fastest variant that fits.  Only LIVE alternatives are selectable (ABI 3
 * dropped the bits that picked superseded kernels: 6 and 13 are ignored).
 *   bit 0       always the generic (global-memory filter) kernels
 *   bit 1       LDS fast path without double-buffering the filter (the integer Barrett kernels)
 *   bit 2       per-pixel threshold compare in the GOP mask kernel even for threshold 0
 *   bit 3       Barrett reductions only (never the FP64 h mod m, which is taken when every filter of a batch has 2^15 <= m < 2^23)
 *   bit 4       run k_hash_table (the pixel-index hash table the insert kernel gathers from, 26 bytes per pixel in an allocation of 32, shared by the
 *               contexts of a process) for every batch instead of once per (device, frame size, seeds).
 *               FOOTPRINT: the table is process-global device memory -- 32 * (width*height + 512) bytes per (device, frame size,
 *               seeds) in use, e.g. 66 MB at 1080p (54 MB used) -- allocated by the first encode of a geometry and released when the last
 *               context holding it is destroyed or moves to another geometry (including one that never uses a table).  Geometries
 *               whose table would exceed 96 MB (2560x1440 and up) never get one: their insert kernels hash the set positions instead.
 *   bit 5       never use that table: the insert kernel hashes the set positions itself
 *   bit 7       filters of several LDS tiles are inserted by the tiled k_insert_tab even inside rbf_encode_gop (default there:
 *               k_insert_positions + k_insert_records)
 *   bits 8..12  temporal chunks of the GOP mask kernel (0 = auto)
 *   bit 14      k_insert_positions hashes the set positions itself whatever the frame size (default: only when the table would
 *               exceed 96 MB)
 *   bit 15      the query kernel never rewrites the hash table (default: a context that is the table's only holder has it
 *               rewritten in every batch, which keeps it in the Infinity Cache)
 *   bits 16..31 LDS tile cap in units of 64 dwords (0 = all of LDS): forces the tiled kernels */
int rbf_ctx_force_generic(rbf_ctx *ctx, int on);
/* Further testing / tuning knobs.  RBF_OPT_SEPARATE_FINISH (0 / 1): 1 = rbf_encode_gop always hands the ones counts out through the
 * separate k_finish_ones launch (default: inside the GOP mask kernel whenever that kernel covers the whole frame).  (Option 1 of ABI 2,
 * the round-2 query kernel, is gone with that kernel: RBF_EINVAL.) */
#define RBF_OPT_SEPARATE_FINISH 2
/* RBF_OPT_INSERT_SLICES (tuning): mask slices (= partial filters) per frame of the single-tile insert kernel; 0 = auto (about one
 * workgroup per CU for the whole batch).  (Options 3, 5 and 99 of ABI 3 / early ABI 4 -- the side-stream compaction, the grouped insert
 * and the kernel-skipping diagnostic -- are gone: RBF_EINVAL.) */
#define RBF_OPT_INSERT_SLICES 4
int rbf_ctx_option(rbf_ctx *ctx, int option, int64_t value);
int rbf_timing_reset(rbf_ctx *ctx);
int rbf_timing_read(rbf_ctx *ctx, int kernel_id, double *total_ms, uint64_t *launches);

/* ---- host-side scalar helpers (no GPU) --------------------------------------------------- */
/* BloomFilterCompressor._calculate_optimal_params with p = ones/n (improved_video_compressor.py:161-196,
 * :211-212).  Returns k=0,l=0 for the reference's "(0, 0)" cases. */
int rbf_optimal_params(uint64_t n, uint64_t ones, double *k, uint64_t *l);
/* floor(k*) and the integer activation threshold T = min{h : RN(h/(2^64-1)) >= k*-floor(k*)}
 * (improved_video_compressor.py:57-58,94-97). */
int rbf_activation_threshold(double k_star, uint32_t *floor_k, uint64_t *threshold);

/* The host step of BloomFilterCompressor.compress for a batch (:211-225): for frame f with
 * ones[f] set bits out of n, fill params[f] (and k[f] if k != NULL).  A frame the reference would
 * NOT Bloom-code (p >= P_STAR, l == 0, or l >= n when guard_l_ge_n != 0) gets params[f].m = 0,
 * which every batch entry point treats as "skip this frame": an empty witness, and a filter row WITHOUT DEFINED CONTENT (the LDS
 * insert kernels leave it alone, the generic path clears every row of the batch). */
int rbf_plan_batch(uint64_t n, const uint64_t *ones, uint32_t nframes, int guard_l_ge_n,
                   rbf_filter_params *params, double *k);

/* ---- A1: residual mask  (VideoFrameCompressor._calculate_frame_diff, :784-808,845) ------- */
/* mask bit = abs_int16(prev - curr) > thr, with numpy's int16 wrap for 16-bit samples; thr is
 * thr_floors[f] for pair f when thr_floors (HOST array of nframes-1 entries) is not NULL -- the
 * adaptive per-frame thresholds of :804-805 -- else thr_floor for every pair.
 * Sample (x, y) of a frame is at base + y*row_pitch_bytes + x*pixel_stride_bytes (so the luma
 * of interleaved YUV444 or a planar plane are both addressable); sample_bytes is 1 or 2.
 * Frame f of the batch is at frames_dev + f*frame_stride_bytes; mask f (f = 0..nframes-2) is the
 * mask between frame f and f+1, written at masks_dev + f*mask_stride_bytes; ones_dev[f] gets
 * np.sum(mask).  mask_stride_bytes must be a multiple of 8 and >= ceil(n/64)*8. */
int rbf_residual_mask_batch(rbf_ctx *ctx, const void *frames_dev, uint64_t frame_stride_bytes,
                            uint32_t nframes, uint32_t width, uint32_t height,
                            uint64_t row_pitch_bytes, uint32_t pixel_stride_bytes,
                            uint32_t sample_bytes, int32_t thr_floor, const int32_t *thr_floors,
                            void *masks_dev, uint64_t mask_stride_bytes, uint64_t *ones_dev);

/* ---- A1, BGR input  (cv2.cvtColor(frame, cv2.COLOR_BGR2GRAY), :794-795) --------------------- */
/* gray = (B*3735 + G*19235 + R*9798 + 2^14) >> 15 per pixel -- OpenCV 4.x's integer path for 8- and
 * 16-bit samples; samples 0,1,2 of a pixel are B,G,R (further channels are ignored).  gray_dev receives
 * nframes dense planes of height*width samples, ready for rbf_residual_mask_batch with
 * pixel_stride_bytes = sample_bytes. */
int rbf_bgr_to_gray_batch(rbf_ctx *ctx, const void *frames_dev, uint64_t frame_stride_bytes, uint32_t nframes,
                          uint32_t width, uint32_t height, uint64_t row_pitch_bytes, uint32_t pixel_stride_bytes,
                          uint32_t sample_bytes, void *gray_dev);

/* ---- A1, planar luma -------------------------------------------------------------------------- */
/* luma_dev[f][y*width + x] = sample 0 of pixel (x, y) of frame f: the dense Y planes of a YUV GOP (the reference reads
 * frame[:, :, 0], improved_video_compressor.py:788-791; its YUVFrame wrapper keeps the same plane as a contiguous copy,
 * fixed_video_compressor.py:292-296).  The residual masks only ever look at luma, so a coder that keeps this block resident
 * (rbf_encode_gop / rbf_residual_mask_batch with pixel_stride_bytes = sample_bytes, frame_stride_bytes = width*height*sample_bytes)
 * moves one third of the bytes of the interleaved frames through the mask stage; the interleaved frames are then only touched
 * by the changed-value gather.  One-time pass at upload. */
int rbf_extract_luma_batch(rbf_ctx *ctx, const void *frames_dev, uint64_t frame_stride_bytes, uint32_t nframes,
                           uint32_t width, uint32_t height, uint64_t row_pitch_bytes, uint32_t pixel_stride_bytes,
                           uint32_t sample_bytes, void *luma_dev);

/* ---- A1, adaptive threshold  (VideoFrameCompressor._estimate_noise_level, :727-744) -------- */
/* For each of nframes luma planes (addressed as above): smoothed = 5x5 median with replicated
 * borders (cv2.medianBlur(frame, 5), :738), noise = frame - smoothed (:741).
 *   moments_dev  out: nframes x 2 int64 -- exact sum(noise) and sum(noise^2)
 *   noise_dev    out, nullable: nframes planes of height*width float32, the array whose float32
 *                np.std the reference takes (:744); every value is an exact integer.
 * The standard deviation, the clamp to [min, max] and the floor that turns it into thr_floors
 * (:756-760) are host float work: engine.py does them from the moments, and falls back to the
 * noise plane when rounding could matter. */
int rbf_noise_moments_batch(rbf_ctx *ctx, const void *frames_dev, uint64_t frame_stride_bytes,
                            uint32_t nframes, uint32_t width, uint32_t height,
                            uint64_t row_pitch_bytes, uint32_t pixel_stride_bytes,
                            uint32_t sample_bytes, int64_t *moments_dev, float *noise_dev);

/* ---- A4 + A5: insert + query/witness  (BloomFilterCompressor.compress loops, :232-253) --- */
/* For each frame f: zero filter f, insert every '1' position of mask f
 * (RationalBloomFilter.add_index, :99-114), then test every position 0..n-1 in order
 * (check_index, :116-138) and append mask bit i to the witness when it passes.
 *   masks_dev     nframes packed masks, stride mask_stride_bytes
 *   params        nframes host structs
 *   filters_dev   out: nframes packed filters, stride filter_stride_bytes >= ceil(m/64)*8
 *   witnesses_dev out: nframes packed witnesses, stride witness_stride_bytes >= ceil(n/64)*8; the first stats[f][RBF_STAT_WITNESS_BITS]
 *                 bits of row f are the witness, zero-padded to whole 64-bit words -- what lies behind them in the row is unspecified
 *   stats_dev     out: nframes x RBF_STATS_PER_FRAME uint64 */
int rbf_bloom_encode_batch(rbf_ctx *ctx, const void *masks_dev, uint64_t mask_stride_bytes,
                           uint64_t n, uint32_t nframes, const rbf_filter_params *params,
                           const rbf_seeds *seeds,
                           void *filters_dev, uint64_t filter_stride_bytes,
                           void *witnesses_dev, uint64_t witness_stride_bytes,
                           uint64_t *stats_dev);

/* ---- A1 + A3 + A4 + A5 for a whole GOP in one call ----------------------------------------- */
/* residual masks of the nframes-1 consecutive pairs -> ones to the host -> rbf_plan_batch ->
 * insert + query.  Blocks once in the middle (the parameter math needs the ones counts on the
 * host); the Bloom kernels are enqueued when it returns.  params_out / k_out (host, nframes-1
 * entries, nullable) receive the per-frame geometry so the caller can size and label the output.
 * filter_stride_bytes must cover every planned filter (0.32*n bits always suffices). */
int rbf_encode_gop(rbf_ctx *ctx, const void *frames_dev, uint64_t frame_stride_bytes,
                   uint32_t nframes, uint32_t width, uint32_t height,
                   uint64_t row_pitch_bytes, uint32_t pixel_stride_bytes,
                   uint32_t sample_bytes, int32_t thr_floor, const int32_t *thr_floors,
                   const rbf_seeds *seeds,
                   void *masks_dev, uint64_t mask_stride_bytes, uint64_t *ones_dev,
                   void *filters_dev, uint64_t filter_stride_bytes,
                   void *witnesses_dev, uint64_t witness_stride_bytes, uint64_t *stats_dev,
                   rbf_filter_params *params_out, double *k_out);

/* The same in two phases (`..._masks()` -> host parameter math -> `..._blooms()`; the reference does these steps one
 * after the other per frame, improved_video_compressor.py:198-266 -- mask, :211-215 parameters, :235-237 insert, :245-253 query).
 *   rbf_encode_gop_begin   checks EVERY argument first (a bad call touches neither the stream nor the caller's buffers), enqueues the
 *                          mask stage (its last workgroup publishes the set-bit counts into pinned host memory and the
 *                          stats rows are cleared on the way) and returns without waiting.
 *   rbf_encode_gop_poll    *ready = 1 once the counts have arrived (never blocks).
 *   rbf_encode_gop_finish  waits for the counts, runs rbf_plan_batch, enqueues insert / reduce / query / compaction; params_out /
 *                          k_out as for rbf_encode_gop.  Clears the pending state even when it fails.
 * One GOP per context may be between begin and finish (a second begin returns RBF_EINVAL); the buffers named in begin must stay
 * valid until the kernels enqueued by finish have run.  A thread that feeds several contexts issues begin(k+1) before finish(k) and
 * never stands still while a mask kernel runs; rbf_encode_gop is begin + finish. */
int rbf_encode_gop_begin(rbf_ctx *ctx, const void *frames_dev, uint64_t frame_stride_bytes,
                         uint32_t nframes, uint32_t width, uint32_t height,
                         uint64_t row_pitch_bytes, uint32_t pixel_stride_bytes,
                         uint32_t sample_bytes, int32_t thr_floor, const int32_t *thr_floors,
                         const rbf_seeds *seeds,
                         void *masks_dev, uint64_t mask_stride_bytes, uint64_t *ones_dev,
                         void *filters_dev, uint64_t filter_stride_bytes,
                         void *witnesses_dev, uint64_t witness_stride_bytes, uint64_t *stats_dev);
int rbf_encode_gop_poll(rbf_ctx *ctx, int *ready);

/* SEVERAL keyframe-delimited runs of one clip in ONE launch sequence (SURVEY.md 8b `rbf_encode_batch ... GOP batching`).  The reference
 * codes frame by frame (improved_video_compressor.py:198-266) and has no GOP loop (:1236-1268); a caller that holds a clip whose frame t
 * is a keyframe iff t % keyframe_interval == 0 hands over up to a few hundred consecutive frames at once and every fixed cost of the
 * launch sequence -- the query kernel's hashing prologue, the mask kernel's ramp and tail, five launches -- is paid once per block
 * instead of once per GOP.
 *   run_starts   HOST array of nframes bytes, nullable (= one run, rbf_encode_gop_begin).  run_starts[t] != 0 for t >= 1: frame t is a
 *                keyframe of the caller's stream, it starts a new run, and PAIR t-1 (frame t against frame t-1) IS NOT CODED: its mask
 *                row is written as zeros, ones_dev[t-1] = 0, its stats are zero (an empty witness), its filter row holds NO DEFINED CONTENT afterwards (the
 *                LDS insert kernels never write it, the generic path clears every row of the batch: read nothing from it),
 *                params_out[t-1] = {m = 0, floor_k = RBF_PAIR_SKIPPED, 0}, k_out[t-1] = 0, and rbf_pack_records gives it a header row
 *                with no payload.  run_starts[0] is ignored.  No frame of a run is read by the mask stage of another run.
 * Everything else as rbf_encode_gop_begin / rbf_encode_gop; rbf_encode_gop_poll / rbf_encode_gop_finish complete either kind of begin.
 * Both begins check EVERY argument before they touch the stream or the caller's buffers.
 * PRECONDITION SINCE ABI 4 (a caller written against ABI 3 that sized the filter rows from its own density bound gets RBF_EINVAL now):
 *   filter_stride_bytes >= rbf_filter_stride_min(width * height), a multiple of 8.
 * The filters are planned in the second half, from counts the first half has not seen yet, so the stride must cover the largest filter
 * the planner can produce for the frame size (l <= 0.31606 n for every density, :181-193) -- ask rbf_filter_stride_min, do not guess. */
#define RBF_PAIR_SKIPPED 0xFFFFFFFFu
uint64_t rbf_filter_stride_min(uint64_t n);
int rbf_encode_runs_begin(rbf_ctx *ctx, const void *frames_dev, uint64_t frame_stride_bytes,
                          uint32_t nframes, uint32_t width, uint32_t height,
                          uint64_t row_pitch_bytes, uint32_t pixel_stride_bytes,
                          uint32_t sample_bytes, int32_t thr_floor, const int32_t *thr_floors,
                          const uint8_t *run_starts, const rbf_seeds *seeds,
                          void *masks_dev, uint64_t mask_stride_bytes, uint64_t *ones_dev,
                          void *filters_dev, uint64_t filter_stride_bytes,
                          void *witnesses_dev, uint64_t witness_stride_bytes, uint64_t *stats_dev);
int rbf_encode_runs(rbf_ctx *ctx, const void *frames_dev, uint64_t frame_stride_bytes,
                    uint32_t nframes, uint32_t width, uint32_t height,
                    uint64_t row_pitch_bytes, uint32_t pixel_stride_bytes,
                    uint32_t sample_bytes, int32_t thr_floor, const int32_t *thr_floors,
                    const uint8_t *run_starts, const rbf_seeds *seeds,
                    void *masks_dev, uint64_t mask_stride_bytes, uint64_t *ones_dev,
                    void *filters_dev, uint64_t filter_stride_bytes,
                    void *witnesses_dev, uint64_t witness_stride_bytes, uint64_t *stats_dev,
                    rbf_filter_params *params_out, double *k_out);
int rbf_encode_gop_finish(rbf_ctx *ctx, rbf_filter_params *params_out, double *k_out);

/* ---- exact-size record of a batch (what the gather to rank 0 moves) ------------- */
/* Compacts the padded output rows of an encode into ONE contiguous self-describing block on the
 * device, without the host learning the witness lengths.  Little-endian uint64 words:
 *   [0] "RBFREC01"  [1] nframes  [2] used bytes  [3] 1 if capacity_bytes was too small (payload incomplete)
 *   per frame 8 words: m, floor_k, threshold, k (float64 bits), witness_bits, filter_ones,
 *                      filter_offset, witness_offset (bytes from the start of the block)
 *   payload, 8-byte aligned rows: filter ceil(m/64)*8 bytes -- or, for a frame the reference does
 *   not Bloom-code (m == 0; :215-225 returns the input itself), its packed mask ceil(n/64)*8 bytes --
 *   and witness ceil(witness_bits/64)*8 bytes.
 * params / k: the host arrays rbf_encode_gop or rbf_plan_batch filled (k nullable -> 0.0).
 * rbf_record_max_bytes: a capacity that can never overflow. */
uint64_t rbf_record_max_bytes(uint32_t nframes, uint64_t n);
int rbf_pack_records(rbf_ctx *ctx, uint32_t nframes, uint64_t n, const rbf_filter_params *params, const double *k,
                     const void *masks_dev, uint64_t mask_stride_bytes,
                     const void *filters_dev, uint64_t filter_stride_bytes,
                     const void *witnesses_dev, uint64_t witness_stride_bytes,
                     const uint64_t *stats_dev, void *record_dev, uint64_t capacity_bytes);

/* ---- A6: decode  (BloomFilterCompressor.decompress loop, :286-307) ------------------------ */
/* out mask bit i = witness[w++] if position i passes the filter, else 0. */
int rbf_bloom_decode_batch(rbf_ctx *ctx, const void *filters_dev, uint64_t filter_stride_bytes,
                           const void *witnesses_dev, uint64_t witness_stride_bytes,
                           uint64_t n, uint32_t nframes, const rbf_filter_params *params,
                           const rbf_seeds *seeds,
                           void *masks_dev, uint64_t mask_stride_bytes);

/* ---- per-index surface  (RationalBloomFilter.add_index / check_index, :99-138) ----------- */
/* filter_dev is a packed filter of params->m bits.  indices_dev: count uint32 indices. */
int rbf_filter_insert_indices(rbf_ctx *ctx, void *filter_dev, const rbf_filter_params *params,
                              const rbf_seeds *seeds, const uint32_t *indices_dev, uint64_t count);
/* out_dev[i] = 1 if indices_dev[i] passes, else 0 (one byte per index). */
int rbf_filter_query_indices(rbf_ctx *ctx, const void *filter_dev, const rbf_filter_params *params,
                             const rbf_seeds *seeds, const uint32_t *indices_dev, uint64_t count,
                             uint8_t *out_dev);

/* ---- string-keyed twins  (rational_bloom_filter.py) ----------------------------------------- */
/* Keys are arbitrary byte strings: key i = keys_dev[offsets_dev[i] .. offsets_dev[i+1]) (count+1
 * uint32 offsets).  standard_k == 0: RationalBloomFilter.add / contains (rational_bloom_filter.py:
 * 139-182; seeds (0, 1, ceil(k*))); standard_k > 0: StandardBloomFilter.add / contains (:29-41), k
 * independent hashes XXH64(key, seed=j) % m (params->m; floor_k / threshold / seeds unused). */
int rbf_filter_insert_keys(rbf_ctx *ctx, void *filter_dev, const rbf_filter_params *params,
                           const rbf_seeds *seeds, uint32_t standard_k,
                           const uint8_t *keys_dev, const uint32_t *offsets_dev, uint64_t count);
int rbf_filter_query_keys(rbf_ctx *ctx, const void *filter_dev, const rbf_filter_params *params,
                          const rbf_seeds *seeds, uint32_t standard_k,
                          const uint8_t *keys_dev, const uint32_t *offsets_dev, uint64_t count,
                          uint8_t *out_dev);

/* ---- A2 / A8: changed-value gather / scatter  (:811-842, :886-903) ------------------------ */
/* Gather, in raster order, the `channels` samples of every pixel whose mask bit is 1 out of
 * an interleaved frame (sample c of pixel (x,y) at base + y*row_pitch + x*pixel_stride + c*sample_bytes)
 * into values_dev (count*channels samples); count_dev gets the number of pixels. */
int rbf_gather_values(rbf_ctx *ctx, const void *frame_dev, uint32_t width, uint32_t height,
                      uint64_t row_pitch_bytes, uint32_t pixel_stride_bytes, uint32_t sample_bytes,
                      uint32_t channels, const void *mask_dev, void *values_dev, uint64_t *count_dev);
/* The same for every pair of a GOP in one call: pair f (f = 0..nframes-2) gathers from frame f+1
 * under mask f; the blocks are concatenated in frame order, pair f starting at pixel offsets_dev[f]
 * (offsets_dev: nframes entries, the last one = total changed pixels), i.e. at sample
 * offsets_dev[f]*channels of values_dev.  Nothing is written past capacity_pixels pixels.
 * uncovered_dev (nullable, nframes-1 entries): pixels whose mask bit is 0 although some channel
 * differs between frame f and f+1 -- changes a luma-only mask cannot carry (:849-909 applies the
 * mask to all channels), which a lossless caller must route to a keyframe. */
int rbf_gather_values_batch(rbf_ctx *ctx, const void *frames_dev, uint64_t frame_stride_bytes, uint32_t nframes,
                            uint32_t width, uint32_t height, uint64_t row_pitch_bytes, uint32_t pixel_stride_bytes,
                            uint32_t sample_bytes, uint32_t channels, const void *masks_dev, uint64_t mask_stride_bytes,
                            void *values_dev, uint64_t capacity_pixels, uint64_t *offsets_dev, uint64_t *uncovered_dev);
/* Inverse: write values back at the mask's '1' pixels (frame_dev is updated in place). */
int rbf_scatter_values(rbf_ctx *ctx, void *frame_dev, uint32_t width, uint32_t height,
                       uint64_t row_pitch_bytes, uint32_t pixel_stride_bytes, uint32_t sample_bytes,
                       uint32_t channels, const void *mask_dev, const void *values_dev);

#ifdef __cplusplus
}
#endif
#endif /* RBF_H */
