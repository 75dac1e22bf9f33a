/*
 * lens_amd.h -- C ABI of liblens_amd.so: the MI355X (gfx950) implementation of the LENS
 * inference-and-matching hot path.  Plain pointers and sizes only; no torch / HIP types in
 * the signatures (a HIP stream is passed as an opaque `void *`, 0 = the null stream).
 *
 * The reference (AdamDHines/LENS) is pure Python and has no FFI; each entry point below names
 * the reference code it replaces (paths relative to the reference repository); a LENS maintainer
 * binds them with ctypes.
 *
 * Conventions
 *   - every `d_*` pointer is DEVICE memory owned by the caller; `h_*` pointers are host memory.
 *     The library never allocates, frees or synchronises; all work is enqueued on `stream`.
 *   - return value: 0 = LENS_OK; negative = argument error (nothing was enqueued);
 *     positive = hipError_t of the failed launch.  Nothing throws across the boundary.
 *   - the entry points keep no state between calls except process-wide caches that are written once and
 *     never change a result: the tuning overrides LENS_AMD_OUTPUT_PPL / LENS_AMD_OUTPUT_WHOLE / LENS_AMD_TRAIN_LPR
 *     (read from the environment at first use), two "kernel attributes checked" flags (atomic), and the per-device
 *     address of the scratch array of lens_forward_counts_static (behind a mutex).  The overrides of the carried-state
 *     layout, LENS_AMD_CARRY_L2 and LENS_AMD_CARRY_SHARED, are read from the environment at every call (tests walk one
 *     shape through every layout); they change lens_event_scale too, so they must not change between a stream's
 *     producer and its consumer.  Calls from several host threads are safe; work on one device is ordered by the streams
 *     the caller passes.
 *
 * Data layouts (row-major, last index fastest)
 *   frames      uint8  [Q][HW]            event-count frames (PNG pixels), HW = roi_dim*roi_dim
 *   sel         int32  [N]                flat pixel index of input neuron n (centre-pixel conv)
 *   codes       uint8  [T][N]             threshold codes: input n spikes at step t iff px >= code
 *   w1t         float  [N][ldw1]          feature_layer.w.weight TRANSPOSED  (weight is [F][N])
 *   w2t         float  [F][ldw2]          output_layer.w.weight TRANSPOSED   (weight is [P][F]);
 *                                         ldw2 >= roundup(P, 256), padding columns zero
 *   events      uint32 [Q][cap]           per-query feature-spike stream, see lens_feature_events
 *   counts      float  [Q][ldo]           output spike counts = rows of the similarity matrix
 */
#ifndef LENS_AMD_H
#define LENS_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LENS_AMD_ABI_VERSION 11

/* Rows of slack behind the currents buffer of lens_feature_currents / lens_feature_fire. */
#define LENS_FIRE_SLACK_ROWS 32

/* Size of the library-owned device array behind lens_forward_counts_static (held once per GPU in use). */
#define LENS_STATIC_SCRATCH_BYTES (64u << 20)

enum {
    LENS_OK = 0,
    LENS_ERR_NULL = -1,        /* a required pointer is NULL */
    LENS_ERR_SIZE = -2,        /* a size / stride argument is out of range */
    LENS_ERR_UNSUPPORTED = -3, /* valid request outside what the kernels implement (see each function) */
    LENS_ERR_ALIGN = -4,       /* a pointer / leading dimension violates the stated alignment */
    LENS_ERR_SCRATCH = -5      /* lens_forward_counts_static: the batch does not fit the static scratch array */
};

/* ABI version of the loaded library (== LENS_AMD_ABI_VERSION it was built with). */
int lens_amd_abi_version(void);

/* Human-readable text for a negative status of this library ("" for others). */
const char *lens_amd_strerror(int status);

/*
 * Event list -> count frames.   Replaces lens/collect_data.py:193-199 `create_images`
 * (twin: lens/tools/manual_eventframe_generator.py:6-11):
 *     frame = zeros(roi, roi, int); frame[e.y-1, e.x-1] += 1 per event; astype(uint8)
 * d_xy        int16 [n_events][2]   (x, y) per event, 1-based like the sensor's
 * d_frame_off int64 [n_frames+1]    events of frame i are [off[i], off[i+1])
 * d_frames    uint8 [n_frames][roi][roi]  (count mod 256, like the uint8 cast)
 * d_first_bad int64 scalar, caller sets it to INT64_MAX; receives the smallest event position whose
 *             (y-1, x-1) is outside [-roi, roi) (torch indexing would raise IndexError there;
 *             indices in [-roi, -1] wrap like Python's).  Frames are unspecified if it fires.
 * roi <= 200 (the histogram lives in LDS); larger -> LENS_ERR_UNSUPPORTED.
 */
int lens_accumulate_events(const int16_t *d_xy, const int64_t *d_frame_off, int n_frames, int roi,
                           uint8_t *d_frames, int64_t *d_first_bad, void *stream);

/*
 * HOST helper: threshold codes of the rate-coding raster.  Replaces the per-item
 *     torch.manual_seed(50); (torch.rand(T, H*W) < img/255).float()
 * of lens/src/dataset.py:23,120-121: with U the (constant) rand tensor,
 *     code[t][n] = min { p in 1..255 : U[t][n] < (float)p / 255.0f }   (255 if none)
 * so that  (U < px/255)  <=>  (px >= code)  for every uint8 px.
 * h_u_sel  float [T][N]  the N selected columns of U (host memory)
 * h_codes  uint8 [T][N]  (host memory)
 */
int lens_build_threshold_codes(const float *h_u_sel, int T, int N, uint8_t *h_codes);

/*
 * HOST helper: for every input neuron n and pixel value p, the number of time steps in which n spikes,
 *     h_active_steps[n][p] = #{ t : h_codes[t][n] <= p },        uint16 [N][256], T <= 65535.
 * The sum over a frame's selected pixels is the exact number of input spikes of that query: the optional load-balancing
 * hint of lens_feature_events.  (No counterpart in the reference, whose loop handles one query at a time.)
 */
int lens_build_active_steps(const uint8_t *h_codes, int T, int N, uint16_t *h_active_steps);

/* Words of `events` a query needs in the worst case: 2 * T * max(F, 1) rounded up to whole 32-word chunks. */
size_t lens_event_capacity_words(int T, int F);

/* Width, in places, of the W2^T tile lens_output_counts stages in LDS for these arguments: 64, 128, 192 or 256
 * (whichever fits F and costs least), or 0 when no tile fits and the weights are read through L2.  `carry` and
 * `n_groups` as passed to lens_output_counts (n_groups <= 0: its default).  With carried state every tile is
 * walked by one wavefront of its own, so the width is chosen for the number of tiles resident at once rather than
 * for the cost per place.  (No counterpart in the reference: a property of this library's kernels.) */
int lens_output_tile_places(int F, int P, int carry, int n_groups);

/* How lens_output_counts lays the work out for these arguments (a property of this library's kernels, no counterpart in
 * the reference): LENS_LAYOUT_BATCH (carry == 0: queries are independent, workgroups of 16 wavefronts share a tile),
 * or, with carried state -- every stream sequential in its query index -- LENS_LAYOUT_LONE_LDS (one wavefront per
 * (64-place tile, stream) with its own LDS copy of the tile), LENS_LAYOUT_LONE_L2 (the same wavefronts reading their rows
 * through L2: all resident at once however large the database) or LENS_LAYOUT_SHARED (the batch mode's wide LDS tile
 * shared by up to 16 streams, one wavefront each: many streams on a large database).  Chosen by a cost model of the
 * measured time per query-step; lens_output_tile_places and lens_event_scale follow the choice. */
#define LENS_LAYOUT_BATCH 0
#define LENS_LAYOUT_LONE_LDS 1
#define LENS_LAYOUT_LONE_L2 2
#define LENS_LAYOUT_SHARED 3
int lens_output_layout(int F, int P, int carry, int n_groups);

/* LENS_LAYOUT_SHARED only, else 0: the number of places at the end of the database that lens_output_counts runs as a
 * second launch of lone wavefronts on 128-place LDS tiles (same streams, same event scale).  The workgroups of a launch
 * all take the same time, so a few wide tiles beyond whole rounds of the CUs would cost a whole round more (100 000
 * places as 16 streams: 521 tiles of 192 = 2.04 rounds); where the cost model says it pays, the wide tiles cover whole
 * rounds and the rest is this tail.  LENS_AMD_CARRY_TAIL=0 switches it off.  (No counterpart in the reference.) */
int lens_output_tail_places(int F, int P, int carry, int n_groups);

/* 1 when lens_output_counts reads W2^T through L2 for these arguments instead of staging tiles in LDS: no tile fits
 * (lens_output_tile_places == 0), or carried state with more (64-place tile, stream) wavefronts than the CUs' LDS surely
 * holds tiles at once -- a carried stream is sequential in the query index, so rounds of LDS-resident tiles would add
 * up, while through L2 every tile's wavefront is resident from the start (100 000 places, one stream: 13.1 -> 5.6 ms per
 * 256 queries).  The tile width stays 64 places; the event scale is then the byte stride of a row of W2^T,
 * 4 * roundup(P, 256) (an event's offset word is its row's byte offset), so lens_output_counts must be given exactly that
 * ldw2 (LENS_ERR_SIZE otherwise) and F * ldw2 * 4 must stay below 2^31 (LDS tiles otherwise).  (No counterpart in the
 * reference.) */
int lens_output_reads_through_l2(int F, int P, int carry, int n_groups);

/* The `ev_scale` lens_feature_events must be given so that lens_output_counts (same F, P, carry and split) can
 * consume its streams: the byte stride of one W2^T row in that tile (512 for the 128- and the 192-place tile,
 * whose third places sit in a plane of their own; 1 if no tile fits and the weights are read through L2; for carried
 * state read through L2 -- lens_output_reads_through_l2 -- the byte stride of a row of W2^T at the minimal ldw2). */
int lens_event_scale(int F, int P, int carry, int n_groups);

/*
 * Frames -> feature-layer spike streams.  Replaces, for a batch of Q queries,
 *   lens/src/dataset.py:99-127        (normalise + rate-code raster, via `codes`)
 *   lens/run_model.py:130-137,141     (centre-pixel Conv2d + IAF #1, an exact identity)
 *   lens/run_model.py:143-144         (feature Linear + sinabs IAF #2), query loop :229-238
 * Arithmetic: I[f] = ascending-n chain of fmaf(x[n], w1[f][n], acc) from +0 (x binary), then the
 * sinabs IAF rule (threshold 1, multi-spike, subtract reset, v_min -1).
 *
 * Queries are split into `n_streams` contiguous runs.  carry != 0: each run is integrated
 * sequentially with the membrane state carried from query to query (the reference never resets;
 * n_streams = 1 is its exact semantics); d_vfeat [n_streams][F] is read at the start and written
 * at the end.  carry == 0: state is zeroed before every query, d_vfeat is not touched (may be
 * NULL), n_streams only sets the parallelism (<= 0: one stream per query).
 *
 * Output stream of query q, d_events + q*cap_words, d_nwords[q] words (a multiple of 32).  The layout is
 * private to this library (lens_output_counts is its only consumer) and may change with the ABI version:
 *   per time step that has spikes, the spiking features in ascending f as two-word events
 *       word 0 = f * ev_scale (bit 31 of the first event of each 16-event chunk: some count in the chunk is not 1)
 *       word 1 = float(spike count); sign bit = LAST event of the step; mantissa LSB (LAST events with a
 *                count < 2^23 only) = the following step(s) have no spikes and are not written at all
 *   a step without spikes that cannot be dropped that way (step 0, or after a count >= 2^23) is the event
 *   {0, -0.0f}; {0, +1.0f} pads to a multiple of 32 words.  Dropping is exact: see lens_kernels.hip.
 * d_featspk (optional, NULL to skip): float [Q][T][F] dense spike counts.
 * d_active_steps (optional, NULL to skip): the table of lens_build_active_steps on the device.  With it, in batch mode
 *        (carry == 0, one stream per query) the 16 queries of a workgroup are dealt to its wavefronts by their number of
 *        input spikes so that the CU's four SIMDs finish together; results do not depend on it.
 * Limits: N <= 1024, F <= 2048, cap_words >= lens_event_capacity_words(T, F), ev_scale * F < 2^31.
 */
int lens_feature_events(const uint8_t *d_frames, int Q, int HW,
                        const int32_t *d_sel, int N,
                        const uint8_t *d_codes, int T,
                        const float *d_w1t, int F, int ldw1,
                        float *d_vfeat, int n_streams, int carry,
                        uint32_t *d_events, int cap_words, uint32_t *d_nwords, int ev_scale,
                        float *d_featspk, const uint16_t *d_active_steps, void *stream);

/*
 * The feature layer for CARRIED state in two passes (same replacement as lens_feature_events with carry != 0, same
 * streams bit for bit, several times faster for few streams): the reference never resets its IAF layers
 * (lens/run_model.py:234-242), which makes a stream sequential in the query index -- but only the integrate-and-fire
 * update is; the input current of a step depends on the frame alone.
 *
 * Pass 1, all queries in parallel.  Replaces lens/src/dataset.py:99-127 (raster), lens/run_model.py:130-137,141
 * (centre-pixel Conv2d + IAF #1) and the Linear of :143:
 *     I[q][t][f] = ascending-n chain of fmaf(x[n], w1[f][n], acc) from +0,
 *     stored with the four steps of a group side by side (lens_feature_fire fetches them with one 16-byte load):
 *     d_currents float [Q][T4 / 4][ldc][4], T4 = T rounded up to a multiple of 4 (steps T..T4-1 are written as zeros),
 *     16-byte aligned; I[q][t][f] sits at ((q * T4 / 4 + t / 4) * ldc + f) * 4 + t % 4.
 *     ldc >= F and a multiple of 64 (columns F..ldc-1 are written as zeros).  N <= 1024, F <= 1024.
 *     lens_feature_fire prefetches ahead: allocate LENS_FIRE_SLACK_ROWS more rows of ldc floats behind the last
 *     query (never written, contents irrelevant).
 */
int lens_feature_currents(const uint8_t *d_frames, int Q, int HW, const int32_t *d_sel, int N,
                          const uint8_t *d_codes, int T, const float *d_w1t, int F, int ldw1,
                          float *d_currents, int ldc, void *stream);

/* Bytes of the `d_scratch` argument of lens_feature_fire (per stream: the spike counts of the steps whose counts exceed 1). */
size_t lens_feature_fire_scratch_bytes(int T, int F, int n_streams);

/*
 * Pass 2, one workgroup per carried stream, sequential in the query index.  Replaces the sinabs IAF #2 of
 * lens/run_model.py:144 over the query loop :229-238: v += I; s = (v > 0) * trunc(v); v -= s; v = relu(v + 1) - 1,
 * and writes the spike streams of lens_feature_events (same layout, same bits).
 * d_vfeat [n_streams][F] carried state, read and written; queries are split into n_streams contiguous runs as in
 * lens_feature_events.  d_scratch: lens_feature_fire_scratch_bytes(T, F, n_streams) bytes, 4-byte aligned, contents
 * irrelevant.  d_currents: as written by lens_feature_currents, with its slack rows.  F <= 768; about T * (ceil(F/64) * 24 + 32)
 * bytes of tables must fit LDS (else LENS_ERR_UNSUPPORTED: use
 * lens_feature_events).
 */
int lens_feature_fire(const float *d_currents, int ldc, int Q, int T, int F,
                      float *d_vfeat, int n_streams,
                      uint32_t *d_events, int cap_words, uint32_t *d_nwords, int ev_scale,
                      void *d_scratch, void *stream);

/*
 * Feature-spike streams -> output spike counts (rows of the similarity matrix).  Replaces
 *   lens/run_model.py:145,155   (output Linear = the place database, + sinabs IAF #3)
 *   lens/run_model.py:239       (spikes.sum(dim=0)),  query loop :229-246
 * Event-driven: I[p] = chain of fmaf(count, w2[p][f], acc) over the step's spiking features in
 * ascending f, which is bit-identical to the dense ascending-f chain.
 *
 * d_w2t: [F][ldw2] with ldw2 % 256 == 0, ldw2 >= P, 16-byte aligned, padding columns zero
 *        (a database shard is simply a column range of this matrix).
 * n_groups / carry: as n_streams / carry of lens_feature_events (the two calls must use the same
 *        split, and the streams must have been written with lens_event_scale of the same F, P, carry and split).
 *        carry != 0: d_vout [n_groups][ldv] carried state, read and written; the work is laid out as
 *        lens_output_layout says (lone wavefronts per tile and stream on LDS tiles or through L2, or wide LDS tiles
 *        shared by 16 streams): same results, bit for bit.
 * d_counts: float [Q][ldo], ldo >= P; columns >= P are not written.
 * cap_words: words between two queries' streams, a multiple of 32 and >= 32 (whole 32-word chunks are read, the
 *        first one of every query before its length is known); d_nwords[q] <= cap_words.
 */
int lens_output_counts(const uint32_t *d_events, int cap_words, const uint32_t *d_nwords, int Q,
                       const float *d_w2t, int F, int P, int ldw2,
                       float *d_vout, int ldv, int n_groups, int carry,
                       float *d_counts, int ldo, void *stream);

/*
 * lens_feature_events followed by lens_output_counts, with the spike-stream workspace (`events`, `nwords`) in
 * a library-owned static device array instead of caller memory.  It has no counterpart in the reference: it
 * exists for callers that cannot allocate device memory -- code under HIP graph capture, tight serving loops,
 * harnesses that forbid hipMalloc.  The call allocates and frees nothing and synchronises nothing; the caller
 * still passes every input and output (arguments as in the two calls above; n_streams / carry apply to both
 * layers, d_vfeat [n_streams][F] and d_vout [n_streams][ldv] are the carried state when carry != 0).
 * Limit: 256-byte-rounded 4*Q + 4*Q*lens_event_capacity_words(T, F) <= LENS_STATIC_SCRATCH_BYTES (64 MiB: 167
 *        queries at T = 250, F = 200), otherwise LENS_ERR_SCRATCH.  Every argument of both layers is checked before the
 *        first kernel is enqueued: a negative status means nothing was enqueued and no carried state has moved.
 * The array's device address is resolved on the first call on each device (hipGetSymbolAddress); make that first
 * call -- Q = 0 is enough -- outside any stream capture.  Every call on a GPU shares that GPU's array, which
 * keeps the previous call's contents (nothing in it has to start at zero: each call writes every word it later
 * reads): calls that overlap on one GPU (two streams, two host threads) race.
 */
int lens_forward_counts_static(const uint8_t *d_frames, int Q, int HW, const int32_t *d_sel, int N,
                               const uint8_t *d_codes, int T, const float *d_w1t, int F, int ldw1,
                               const float *d_w2t, int P, int ldw2,
                               float *d_vfeat, float *d_vout, int ldv, int n_streams, int carry,
                               float *d_counts, int ldo, void *stream);

/*
 * Sequence matcher, full matrix.  Replaces lens/run_model.py:248-252:
 *     S = (conv2d(out[1,1,Q,P], eye(L)) / L).T
 * but stores it query-major: d_S [Q-L+1][lds], d_S[i][j] = (1/L) * sum_{d<L} counts[i+d][j+d],
 * j < P-L+1 (the caller transposes for the reference's [place][query] orientation).  L >= 1.
 */
int lens_seqmatch(const float *d_counts, int Q, int P, int ldo, int L, float *d_S, int lds, void *stream);

/*
 * Sequence matcher fused with per-query top-K.  Replaces lens/run_model.py:248-252 followed by
 * the `S.argsort(0)[-K:]` of lens/src/metrics.py:219, without materialising S.
 * For query column i < Q-L+1: the K places j < n_places with the largest S[i][j]; ties broken
 * towards the LARGER place index (what a stable ascending argsort + [-K:] returns).
 * n_places: number of sequence-matched places this call ranks (P-L+1 on one GPU; for a database
 *           shard that carries L-1 halo columns, its own place count).
 * place_offset is added to the reported indices (global index of the shard's first place).
 * d_topk_idx int32 [Q-L+1][K], d_topk_val float [Q-L+1][K], best first; missing entries (-1, -inf).
 * 1 <= K <= 64, L >= 1.
 */
int lens_seqmatch_topk(const float *d_counts, int Q, int ldo, int n_places, int L, int K, int place_offset,
                       int32_t *d_topk_idx, float *d_topk_val, void *stream);

/*
 * Merge R per-shard top-K lists (e.g. after an RCCL all-gather) into one, same ordering rule.
 * d_idx_in / d_val_in: [R][Qs][K];  d_idx_out / d_val_out: [Qs][K].
 */
int lens_topk_merge(const int32_t *d_idx_in, const float *d_val_in, int R, int Qs, int K,
                    int32_t *d_idx_out, float *d_val_out, void *stream);


/* ---------------------------------------------------------------------------------------------
 * Training path (`python main.py --train_model`), the reference's training run (benchmark size: a 100 000-row output layer).
 * ------------------------------------------------------------------------------------------- */

/* Padded fan-in the training kernel uses for n_in inputs (64/128/200/208/256/512/1024/2048; 0 = n_in > 2048, unsupported).
 * x_table / mpre_table rows must be zero padded to a leading dimension >= this value. */
int lens_train_in_pad(int n_in);

/*
 * Frozen-layer pass for training a later layer.  Replaces lens/train_model.py:163-168
 * (`spikes = layer.w(spikes); spikes = clamp_spikes(spikes, layer)`, lens/src/blitnet.py:177-181) for all
 * S samples at once:  y[s][o] = clamp(sum_n x[s][n]*w[o][n] - thr[o], 0, 1), Linear = ascending fmaf chain.
 * d_x [S][ldx], d_w [n_out][ldw] (state_dict layout), d_thr [n_out], d_y [S][ldy].
 */
int lens_layer_forward_clamped(const float *d_x, int S, int ldx, const float *d_w, int ldw, const float *d_thr,
                               int n_in, int n_out, float *d_y, int ldy, void *stream);

/*
 * T consecutive training steps of one layer, in place.  Replaces the body of LENS_Trainer.train_layer
 * (lens/train_model.py:157-181) and calc_stdp (lens/src/blitnet.py:183-254): forward, clamp, STDP
 * (normal rule, or the spike-forcing rule when d_force_idx != NULL), sign-masked weight clamps, ITP
 * threshold update and inhibitory homeostasis, each tensor expression in its written order in f32.
 *   d_w [n_out][ldw], d_thr [n_out]                  in/out: <layer>.w.weight rows ZERO PADDED to ldw >= lens_train_in_pad(n_in)
 *                                                    (ldw % 4 == 0, 16-byte aligned; the padding is written back as zeros), <layer>.thr
 *   d_exc, d_inh uint8 [n_out][ldm]                  havconnCombinedExc / Inh (fixed at construction), rows zero padded, ldm >= in_pad
 *   d_fire_rate [n_out], have_rate                   layer.fire_rate, layer.have_rate
 *   d_x_table [S][ldx]                               the layer's input vector per sample (zero padded rows)
 *   d_mpre_table [S][ldx] (spike forcing only)       x_table / prev_layer.fire_rate   (blitnet.py:201-205)
 *   d_order int32 [T]                                sample presented at each step
 *   d_force_idx int32 [T] or NULL                    int(labels / filter) per step (spike forcing = output layer)
 *   d_eta_ip, d_eta_stdp float [T]                   learning rates in effect at each step (annealing schedule)
 * n_in <= 2048 (the weight row of a neuron lives in the registers of 1, 2, 4, 8 or 16 adjacent lanes).
 */
int lens_train_layer(float *d_w, int ldw, float *d_thr, const uint8_t *d_exc, const uint8_t *d_inh, int ldm,
                     const float *d_fire_rate, int have_rate, const float *d_x_table, const float *d_mpre_table, int ldx,
                     const int32_t *d_order, const int32_t *d_force_idx, const float *d_eta_ip, const float *d_eta_stdp,
                     int n_out, int n_in, int T, void *stream);


/*
 * Sum-of-absolute-differences baseline (`--sad`).  Replaces `torch.cdist(a, b, 1)` of lens/src/sad.py:34-38
 * on the raw frames: d_dist[i][j] = sum_k |query[i][k] - reference[j][k]|  (exact integer, stored as float32).
 * d_query uint8 [Q][HW], d_reference uint8 [P][HW], HW % 16 == 0, 16-byte aligned, 8*HW <= 160 KiB (LDS tile).
 */
int lens_sad_matrix(const uint8_t *d_query, int Q, const uint8_t *d_reference, int P, int HW, float *d_dist, int ldo,
                    void *stream);


/*
 * DENSE variant of lens_output_counts on the f32 matrix cores (v_mfma_f32_32x32x2_f32), state reset per query.
 * Same replacement (lens/run_model.py:145,155,239) and bit-identical results (the f32 MFMA is a k-ordered fmaf
 * chain); kept as the measured "descriptor x database GEMM" comparison -- it is ~8x slower than the
 * event-driven kernel because it cannot skip the ~97% zero spikes.
 * d_featspk float [Q][T][ld_f] dense feature spike counts (lens_feature_events' optional output), F <= 256.
 */
int lens_output_counts_dense(const float *d_featspk, int Q, int T, int F, int ld_f, const float *d_w2t, int P, int ldw2,
                             float *d_counts, int ldo, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* LENS_AMD_H */
