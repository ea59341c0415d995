/* libmdbn_hip.so -- C-ABI of the MI355X (gfx950) CD-k engine.
 *
 * The reference (glgerard/MDBN, pure Python on Theano) has no FFI: its device boundary is
 * the call of a Theano-compiled step function built from RBM.get_cost_updates
 * (src/rbm.py:258-376, compiled at src/dbn.py:302-312 and src/rbm.py:533-544).  Each entry
 * point below replaces the piece of that compiled graph named in its comment; the Python
 * classes in mdbn_amd/ (same names and signatures as src/rbm.py, src/dbn.py, src/mlp.py)
 * are the only callers, through ctypes (mdbn_amd/_lib.py).
 *
 * Conventions
 *  - every function returns 0 (MDBN_OK) or a negative MDBN_E* code; text via mdbn_last_error
 *  - device buffers are owned by the caller (torch tensors): raw pointers, float32, row-major,
 *    leading dimensions `ld*` counted in floats, ld % 4 == 0 and 16-byte aligned bases
 *    (mdbn_padded_ld gives the recommended ld)
 *  - `stream` is a hipStream_t passed as void*; calls enqueue work and never synchronise
 *  - W is [V, ldh] (n_visible rows, n_hidden columns), as src/rbm.py:104
 *  - random matrices are addressed by (seed, stream_id, step, draw, row_offset): see
 *    mdbn_amd/csrc/philox.h (Philox4x32-10; CPU twins in oracle/philox_np.py, philox_ref.c)
 */
#ifndef MDBN_HIP_H
#define MDBN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MDBN_OK       0
#define MDBN_EINVAL  (-1)   /* bad argument (shape, alignment, null pointer) */
#define MDBN_EHIP    (-2)   /* a HIP runtime call failed */
#define MDBN_ENOSPC  (-3)   /* workspace too small */

/* 2 (round 4): mdbn_cd_args and mdbn_update_args begin with `struct_size` -- a caller built against another layout of
 * either struct is refused with MDBN_EINVAL instead of having trailing fields read from whatever follows its struct. */
#define MDBN_VERSION 2

typedef struct mdbn_ctx mdbn_ctx;

/* Addresses one random matrix (uniform or normal) of one step. */
typedef struct mdbn_rng {
    uint64_t seed;        /* Philox key low/high words                                  */
    uint32_t stream_id;   /* one per RBM layer, xor-ed into the key's high word          */
    uint32_t step;        /* counter of step-function / sampling calls                   */
    uint32_t draw;        /* which random matrix inside the step (0 = U_h0, 2t-1, 2t)    */
    uint32_t reserved;
    uint64_t row_offset;  /* global row of local row 0 (data-parallel shard offset)      */
} mdbn_rng;

/* Everything one CD-k / PCD-k step needs: the compiled step function of
 * src/rbm.py:258-376 (+ the minibatch gather of src/dbn.py:307 / src/rbm.py:538). */
typedef struct mdbn_cd_args {
    uint64_t     struct_size; /* = sizeof(mdbn_cd_args) of the header the caller was built with; anything else: MDBN_EINVAL */
    /* data */
    const float *data;        /* [n_data, ldv] training matrix (train_set_x)                    */
    int64_t      n_data;
    const void  *indexes;     /* [B] minibatch row indices (device), or NULL: rows 0..B-1 of data */
    int32_t      index_is_64; /* 1 = int64 (dbn.py:271), 0 = int32 (rbm.py:528)                  */
    int32_t      gauss;       /* 1 = GRBM (rbm.py:631-699), 0 = Bernoulli RBM                    */
    int32_t      add_noise;   /* GRBM only: 1 = error_free False (rbm.py:652-658)                */
    int32_t      sample_stats;/* 1 = negative visible statistics from nv_SAMPLE, not nv_mean: the
                               * chain_end of compute_symbolic_grad (rbm.py:339-342,378-390)      */
    int32_t      keep_f32;    /* plane path and one-launch path (LDS-resident layers): 1 = also store the float32 copies of ph_mean, -nh_mean (P2), nv_mean
                               * (rows B.. of V2) and the chain samples (hs, vs) that only an inspecting caller reads;
                               * 0 = planes only (the statistics are always written; V2 / P2 / hs / vs are then scratch without defined content).  The
                               * f32-operand path always writes them; trace_h / trace_v imply 1                      */
    int32_t      k;           /* Gibbs steps                                                     */
    int64_t      B, V, H;     /* local minibatch rows, n_visible, n_hidden                       */
    int64_t      ldv, ldh;    /* leading dims of [.,V] and [.,H] matrices (W uses ldh)           */
    /* parameters */
    const float *W, *hbias, *vbias;
    /* PCD: persistent chain [B, ldh] read as chain start and overwritten with nh_sample
     * (rbm.py:308-311,369); NULL = CD */
    float       *persistent;
    /* scratch owned by the caller */
    float       *V2;          /* [2B, ldv]: rows 0..B-1 = v0, rows B..2B-1 = nv_mean (last step) */
    float       *P2;          /* [2B, ldh]: rows 0..B-1 = ph_mean, rows B..2B-1 = -nh_mean       */
    float       *hs;          /* [B, ldh] hidden chain state                                     */
    float       *vs;          /* [B, ldv] visible sample (RBM; GRBM with noise), may be NULL     */
    float       *stats;       /* packed [V*ldh | ldh | ldv | 4]: S, s_h, s_v, cost_sum           */
    void        *workspace;   /* >= mdbn_workspace_bytes(B, V, H)                                */
    int64_t      workspace_bytes;
    mdbn_rng     rng;         /* .draw is ignored (the step numbers its own draws)               */
    /* optional monitoring taps on the Gibbs chain (NULL = off; the role MonitorMode plays at
     * dbn.py:297-300,538-548): every sample the chain feeds onward is also copied here, so a checker can
     * follow the device's chain half-step by half-step */
    float       *trace_h;     /* [k+1][B][ldh]: slot 0 = positive-phase sample, slot t = hidden sample of
                               * Gibbs step t (slot k only when that sample is materialised: PCD)      */
    float       *trace_v;     /* [k][B][ldv]: slot t-1 = Bernoulli visible sample of step t (RBM only)   */
    /* bf16 plane path (optional).  With both pointers set and a shape made of whole 128-row/column tiles
     * (B, V, H multiples of 128, ldv == V, ldh == H, CD without persistent chain) every tensor of the step is
     * split ONCE into three bf16 planes where it is produced and the GEMMs copy plane tiles into LDS by LDS-DMA
     * (csrc/mdbn_planes.hip); otherwise the GEMMs split their f32 operands themselves.  Same arithmetic, same
     * results up to fp32 summation order. */
    void        *planes;      /* scratch of >= mdbn_planes_bytes(B, ldv, ldh) bytes, 16-byte aligned          */
    int64_t      planes_bytes;
    void        *W_planes;    /* [3][V][ldh] bf16: the exact split of W (W = p1 + p2 + p3)                    */
    int32_t      W_planes_valid; /* 1: W_planes already hold the split of the current W; 0: split W first.
                               * mdbn_cd_train_step / mdbn_apply_update (with its W_planes set) keep them in
                               * step with W, so a caller passes 0 only after writing W itself            */
    int32_t      comm_cus;       /* data-parallel mode: CUs left to a collective that runs beside this step; > 0: the plane
                               * GEMMs are launched balanced on (CUs - comm_cus) workgroups instead of one workgroup
                               * per CU (a collective's kernel takes whole CUs, and a full grid on fewer CUs needs a
                               * second round).  0: one workgroup per CU.  mdbn_set_option("comm_cus") overrides 0.  */
    /* Gather-ahead (optional; single-device plane path of mdbn_cd_train_step).  A caller that knows the NEXT minibatch
     * (the trainers do: the epoch's order is drawn up front, src/dbn.py:446-458) passes its indices and a second X2-plane
     * buffer: the statistics kernel's loader waves then gather those rows into the other buffer beside the MFMA main loop,
     * and the next call starts without a gather launch (x_buffer flipped, v0_ready = 1).  Same planes as the gather kernel
     * writes: results are bit-identical with and without.  The library decides per call whether it does it (plane path,
     * fused early update, keep_f32 = 0, <= 4 rows per workgroup) and reports through *ahead_done.
     * Thin-batch path (B <= 32): the update kernel, which streams W once anyway, also gathers the next minibatch into rows
     * 0..B-1 of V2 and leaves the partials of its positive phase x' W' in planes_alt (>= mdbn_ahead_bytes_ctx bytes): the
     * next call (v0_ready = 1; same data, index list and parameters, nothing else run on these buffers in between)
     * starts at its first activation kernel.  Same products in the same order: bit-identical with and without. */
    const void  *next_indexes;   /* [B] indices of the next minibatch (device, same type as indexes), or NULL         */
    void        *planes_alt;     /* second [3][2B][ldv] bf16 X2-plane buffer (mdbn_ahead_bytes_ctx), or NULL            */
    int32_t      x_buffer;       /* X2 planes of THIS step: 0 = inside `planes`, 1 = `planes_alt`                      */
    int32_t      v0_ready;       /* 1: rows 0..B-1 of that buffer already hold this minibatch (gathered ahead)         */
    int32_t     *ahead_done;     /* host pointer (nullable): set to 1 when this call gathered next_indexes ahead, else 0 */
} mdbn_cd_args;

/* Parameter update of src/rbm.py:347-365 from (all-reduced) statistics. */
typedef struct mdbn_update_args {
    uint64_t struct_size;           /* = sizeof(mdbn_update_args); anything else: MDBN_EINVAL */
    float *W, *W_speed;             /* [V, ldh] */
    const float *W0;                /* frozen weight-cost snapshot (rbm.py:415) or NULL = live W */
    float *hbias, *hbias_speed;     /* [H] */
    float *vbias, *vbias_speed;     /* [V] */
    int64_t V, H, ldv, ldh;
    const float *stats;             /* packed as mdbn_cd_args.stats (summed over ranks)          */
    float lr, lambda_1, lambda_2, weightcost, momentum;
    float batch_size;               /* divisor of S: the batch_size ARGUMENT (rbm.py:413)        */
    float n_rows;                   /* divisor of s_h, s_v: rows actually present (rbm.py:416-417) */
    float cost_scale;               /* monitoring cost = stats.cost_sum * cost_scale ...          */
    float *cost_out;                /* ... written here (device scalar) if not NULL               */
    void *W_planes;                 /* NULL, or [3][V][ldh] bf16 planes rewritten with the split of the new W
                                     * whenever this call changes W (phases 0, 2, 3)                          */
    int32_t phase;                  /* 0 = whole rule; 1 = speeds (+cost) only; 2 = parameters only;
                                     * 3 = 1 then 2 in one pass (parameters from the NEW speeds).
                                     * Because the parameter step uses the OLD speed (rbm.py:364-365),
                                     * theta(t+1) never depends on step t's gradient: a data-parallel
                                     * run applies phase 2 at once and phase 1 when the all-reduced
                                     * statistics arrive, overlapping the collective with the next step
                                     * (requires lambda_1 == 0 and weightcost == 0 or a frozen W0). */
    int32_t reserved;
} mdbn_update_args;

int  mdbn_version(void);
int  mdbn_last_error(char *buf, size_t n);
/* SHA-256 (hex) of the sources this library was built from (mdbn_amd/build.py compares it with the
 * sources on disk: a stale prebuilt library is rebuilt or refused, never silently used). */
int  mdbn_source_hash(char *buf, size_t n);

int  mdbn_ctx_create(mdbn_ctx **out, int device);
int  mdbn_ctx_destroy(mdbn_ctx *ctx);

/* Tuning knobs, PER CONTEXT (round 5): mdbn_set_option(ctx, ...) changes what calls made on `ctx` launch and nothing
 * else -- two contexts of one process (two engines, modality-parallel placement in-process) keep their own settings; the
 * context-free sizing calls (mdbn_workspace_bytes, mdbn_planes_eligible) answer for a context with default options, their
 * _ctx forms for the given one.  Results never change beyond fp32 summation order.
 * "gemm_bf16x6" (default 3): bit 0 = statistics GEMM, bit 1 = forward GEMMs run on the bf16 matrix pipe
 *   with exactly split f32 operands and f32 accumulation (f32-grade results) when the problem is made of
 *   whole 128x128 tiles; 0 = always the exact-f32 MFMA kernel.
 * "x6_min_jobs" (default 48): problems with fewer 128x128-tile jobs keep the exact-f32 kernel.
 * "x6_producer_waves" (default 4): producer waves per operand of the bf16x6 kernel (2 | 4).
 * "gemm_bk": GEMM slice depth, 0 = auto, 32 or 64.  "gemm_cw": MFMA waves per SIMD of the tiled
 *   GEMM, 0 = auto (2 for <= 512 rows), 1 or 2.
 * "epilogue_cw": columns per thread of the activation epilogue, 0 = auto, 1, 2 or 4.
 * "fused_epilogue" (default 1): GEMMs that need no split-K apply bias + activation + sampling to
 *   their own output tile instead of writing slabs for a second kernel (bitwise the same result).
 * "fused_update" (default 1): mdbn_cd_train_step applies the weight update inside the statistics
 *   GEMM when that GEMM is not split (and lets the update kernel sum the split-K slabs when it
 *   is); the S block of a->stats is then NOT materialised (s_h, s_v and cost_sum are).  Set 0 to
 *   get S (bitwise the same parameters either way).
 * "fused_finalize" (default 1): with fused_update, the bias statistics / cost / bias update run
 *   inside the statistics GEMM as well (no finalize launch).
 * "skinny_gemm" (default 1): GEMMs of <= 64 output rows, and tiny GEMMs at any row count, use the
 *   register-streaming kernel (no LDS staging); "skinny_fused_max_k" (default 1024) largest K one
 *   block streams alone, "skinny_max_macs" (default 32 Mi) size limit above 64 rows.
 * "stream_x6" (default 2): mid-size passes at more than 64 rows (M * N * K <= "stream_max_macs", default 2^30: the
 *   layers 1024 -> 256 and 2048 -> 400 of BASELINE configs 4 / 5 at B = 512) run UNSPLIT on 32 x 32 / 64 x 32 tiles of the
 *   register-streaming kernel on the bf16 matrix pipe (f32 operands split in registers into their three exact bf16
 *   pieces; six piece products, three when the row operand holds 0/1 samples) with the activation / update epilogue on
 *   the tile: one launch per pass instead of split-K GEMM + slabs + epilogue launch.  2: the small-layer passes of
 *   "skinny_gemm" above 64 rows too; 1: only the passes the LDS-tiled kernels served; 0: off.  "stream_mi": 0 = auto, 1 | 2 = 32-row blocks per tile.
 * "gemm_planes" (default 1): use the bf16 plane path of mdbn_cd_args when its buffers are given and the
 *   shape qualifies.  "planes_mfma" (default 16): its MFMA shape, 16 = v_mfma_f32_16x16x32_bf16, 32 =
 *   v_mfma_f32_32x32x16_bf16 (bit-identical to the f32-operand path). 
 *  "planes_min_work" (default
 *   2^30): smallest B * V * H the plane path serves (smaller whole-tile layers are faster on the f32-operand kernels).
 * "early_w" (default 1): in the plane statistics GEMM with the fused update the parameter half (W and its planes: it
 *   needs only the old W and the old speed, rbm.py:364-365) is applied by the kernel's loader waves during the main loop,
 *   the epilogue only forms the new speed; bitwise the same parameters and speeds (needs lambda_1 == 0 and weightcost == 0
 *   or a frozen W0, else the whole rule stays in the epilogue).
 * "narrow_tiles" (default 1): a forward pass of the plane path whose 128 x 128 plan would split K exactly two ways
 *   (propdown at the c2 shape) runs unsplit on 128 x 64 tiles with the activation fused into the GEMM instead: no slabs,
 *   no epilogue launch; another fp32 summation grouping (one K-long chain instead of two halves).
 * "gather_ahead" (default 1): honour mdbn_cd_args.next_indexes (the next minibatch gathered inside the statistics
 *   kernel); 0 = every step launches its own gather.
 * "comm_cus" (default 0): CUs left to a collective that runs beside the step when mdbn_cd_args.comm_cus is 0 (see
 *   there); "bal_blocks" (tests): the number of workgroups of a balanced launch itself.
 * "epilogue_threads": threads per block of the activation epilogue launch, 0 = auto, 64, 128 or 256.
 * "bf16_inputs" (default 0): REPORTING mode, not a parity path: the plane GEMMs use only the leading bf16 piece of
 *   every operand (one product instead of six; probabilities off by ~4e-3).
 * "update_overlap": 1 = mdbn_cd_train_step overlaps part of the update with the statistics GEMM
 * on a side stream (default 0: measured slower, see csrc/mdbn_capi.hip).
 * "thin_fused" (default 1): minibatches of <= 32 rows (the reference's batch_size = 20, MDBN.py:46) on layers that are not
 *   LDS-resident (ldh <= 512) run as a stream over W (csrc/mdbn_thin.hip): the positive phase and each gibbs_hvh read W
 *   once (propdown and the following propup share one read: a workgroup owns whole rows of W), the rank-2B statistics are
 *   formed in registers inside the update pass -- W is read 2 + k times and written once per CD-k step.  0: the
 *   register-streaming GEMM path.
 * "gchain" (default 0, opt-in): mid-size layers at B > 32 whose W fits the LDS of a group of 2-16 CUs (256 -> 200, 1024 -> 256)
 *   run gather + positive phase + the whole Gibbs chain in ONE launch (csrc/mdbn_gchain.hip): 32-row slabs, each member of a
 *   group holds a block of W's rows in LDS, one in-launch exchange of the upward partials per pass (agent-scope write-through
 *   stores, a flag per member, bounded polls).  Same outputs as the multi-launch path, bit-reproducible; measured slower
 *   than it (an exchange costs what a kernel boundary costs), hence off.
 * "small_fused" (default 1): a layer whose W fits one CU's LDS (V, H <= 512, W image + row buffers <= 160 KB: 512 -> 40,
 *   400 -> 40, 200 -> 20, 100 -> 128, 100 -> 24 -> 3) runs the whole CD-k chain in ONE launch per step -- W staged once, each
 *   workgroup the whole chain for 4-row slabs on v_mfma_f32_4x4x1 out of LDS, partial statistics per workgroup -- plus a
 *   small finish launch (sum of the partials in a fixed order + update); same Philox addressing as every path.  0: the
 *   multi-launch path.  "small_fin_lanes" (0 = by the number of partials | 1, 2, 4, 8, 16): threads of the finish launch
 *   that share one sum. */
/* Introspection of the balanced launches of the data-parallel mode (no GPU work): segment k of workgroup w when
 * `workgroups` workgroups share `tiles` x `stages` evenly.  out[6] = {tile, first stage, end stage, segments of this
 * workgroup, stages of this workgroup, slabs (= sharing workgroups) of that tile}; tile = -1 when k is out of range. */
int  mdbn_bal_segment(int32_t tiles, int32_t stages, int32_t workgroups, int32_t w, int32_t k, int32_t *out);
int  mdbn_set_option(mdbn_ctx *ctx, const char *name, int64_t value);

/* Measurement hook (bench.py): while enabled, every GEMM launch (the dominant kernel) is
 * bracketed by HIP events on its stream; _read synchronises on them and returns the number
 * of launches recorded since enabling and the sum of their durations. */
int  mdbn_kernel_timing(mdbn_ctx *ctx, int enable);
int  mdbn_kernel_timing_read(mdbn_ctx *ctx, int64_t *n_launches, double *total_ms);
/* Per recorded launch (up to `cap`): duration, algorithmic FLOPs 2*M*N*K, FLOPs issued on the matrix
 * pipe the kernel runs on (x6 / x3 for the split-operand bf16 kernels), and
 * kind = 1000*family (0 LDS-tiled, f32 operands; 1 register-streaming; 2 bf16 planes)
 *        + 100*pipe (0 exact-f32 MFMA, 1 bf16 six products, 2 bf16 three products, 3 bf16 one product)
 *        + 10*fused + 2*la + lb  ((la, lb): 1 = propup, 0 = propdown, 3 = statistics).
 * *n_launches receives the number recorded (may exceed cap). */
int  mdbn_kernel_timing_detail(mdbn_ctx *ctx, int64_t cap, double *ms, double *alg_flop,
                               double *pipe_flop, int32_t *kind, int64_t *n_launches);

/* bytes of split-K / reduction scratch the calls below need for shapes up to (B, V, H) */
int  mdbn_workspace_bytes(int64_t B, int64_t V, int64_t H, int64_t *bytes);
/* ... under the options of `ctx` (the plans, hence the scratch, depend on them) */
int  mdbn_workspace_bytes_ctx(mdbn_ctx *ctx, int64_t B, int64_t V, int64_t H, int64_t *bytes);
/* leading dimension this library recommends for a [., cols] matrix (currently round_up(cols, 4);
 * see the measurement note in csrc/mdbn_capi.hip).  Any ld % 4 == 0, ld >= cols is accepted. */
int  mdbn_padded_ld(int64_t cols, int64_t *ld);
/* bytes of plane scratch (mdbn_cd_args.planes) for a minibatch of B rows: planes of [v0; nv], [ph; -nh],
 * the hidden and the visible sample */
int  mdbn_planes_bytes(int64_t B, int64_t ldv, int64_t ldh, int64_t *bytes);
/* bytes of the second X2-plane buffer of the gather-ahead (mdbn_cd_args.planes_alt) */
int  mdbn_planes_alt_bytes(int64_t B, int64_t ldv, int64_t *bytes);
/* bytes of mdbn_cd_args.planes_alt for this shape, whichever path serves it (plane path: the same as
 * mdbn_planes_alt_bytes; thin-batch path: the partials of the next step's positive phase) */
int  mdbn_ahead_bytes_ctx(mdbn_ctx *ctx, int64_t B, int64_t V, int64_t H, int64_t ldv, int64_t ldh, int64_t *bytes);
/* Does the plane path of mdbn_cd_step / mdbn_cd_train_step serve this shape under the CURRENT options (B and V whole
 * 128-row / 128-column tiles, ldv == V; the hidden side on a leading dimension of whole tiles, ldh % 128 == 0 and
 * H <= ldh < H + 128 -- a ragged hidden width rides on zero pad columns --; "gemm_planes" on, B * V * ldh >=
 * "planes_min_work")?  The one statement of the rule: a host allocates mdbn_cd_args.planes / W_planes for exactly these shapes.  (Per call the step also needs CD
 * without a persistent chain, no sample_stats and no GRBM noise.)  W planes handed to a step with W_planes_valid = 0
 * are re-split on entry whichever path the step takes, so they are valid after ANY step that received them. */
int  mdbn_planes_eligible(int64_t B, int64_t V, int64_t H, int64_t ldv, int64_t ldh, int32_t *eligible);
int  mdbn_planes_eligible_ctx(mdbn_ctx *ctx, int64_t B, int64_t V, int64_t H, int64_t ldv, int64_t ldh, int32_t *eligible);
/* exact three-way bf16 split of an f32 matrix [rows, ld] into planes [3][rows][ld] (x = p1 + p2 + p3) */
int  mdbn_split_planes(mdbn_ctx *ctx, void *stream, const float *x, int64_t rows, int64_t ld, void *planes);
/* floats in the packed statistics buffer [S (V*ldh) | s_h (ldh) | s_v (ldv) | 4] */
int  mdbn_stats_floats(int64_t V, int64_t ldv, int64_t ldh, int64_t *n);

/* minibatch gather: train_set_x[indexes]  (src/dbn.py:307, src/rbm.py:538) */
int  mdbn_gather_rows(mdbn_ctx *ctx, void *stream, const float *src, int64_t n_rows,
                      int64_t cols, int64_t ld_src, const void *indexes, int index_is_64,
                      int64_t n_idx, float *dst, int64_t ld_dst);

/* The same gather for a source in PINNED HOST memory (device-accessible: hipHostMalloc / torch pin_memory), read over
 * PCIe by a small-footprint kernel: `workgroups` (0 = 32) workgroups of `threads` (0 = 256; 64 | 128 | 256) threads walk
 * the rows, four 16-byte loads in flight per thread, 20 VGPRs -- the streamed form of src/utils.py:113-115's table for
 * data that is not uploaded whole; meant for a side stream, one minibatch ahead of the step.  dst is device memory. */
int  mdbn_gather_rows_host(mdbn_ctx *ctx, void *stream, const float *src, int64_t n_rows,
                           int64_t cols, int64_t ld_src, const void *indexes, int index_is_64,
                           int64_t n_idx, float *dst, int64_t ld_dst, int workgroups, int threads);

/* out[r] = table[indexes[r]] on the HOST, by `threads` CPU threads (row-sized memcpys; indexes NULL = the first n rows):
 * the host half of the row feeder below, callable on its own (no GPU needed).  MDBN_EINVAL on an index outside
 * [0, n_rows). */
int  mdbn_host_gather_rows(const float *table, int64_t n_rows, int64_t cols, int64_t ld,
                           const int64_t *indexes, int64_t n, float *out, int64_t ld_out, int threads);

/* ROW FEEDER of a host-resident training table (the streamed form of src/utils.py:113-115's table for data that is not
 * uploaded whole) that keeps the CUs out of it: `threads` CPU threads gather a minibatch's rows into a pinned staging
 * slot, ONE hipMemcpyAsync moves the slot to the device on the feeder's own copy stream (SDMA), the consuming stream
 * waits for that copy's event.  Unlike mdbn_gather_rows_host's kernel, which slows every GEMM it runs beside, the copy
 * costs the step nothing; the ring of `slots` (>= 2; 3 keeps two minibatches in flight beside the one being read) hides
 * the gather and the PCIe time behind the steps.
 *   create : table [n_rows, ld] in host memory (pinned or pageable), device_slots[slots] device buffers of
 *            [max_rows, ld_device] floats owned by the caller (pad columns are written as zero);
 *   submit : queue a minibatch (the indexes are copied; NULL = rows 0 .. n-1); returns at once with a ticket;
 *   acquire: tickets in submission order; blocks the HOST until the ticket's copy has been enqueued, makes `stream` wait
 *            for it, returns the slot whose device buffer holds the rows;
 *   release: the work that reads the slot has been enqueued on `stream`: the slot may be overwritten after it;
 *   cancel : forget everything submitted and not acquired.
 * Thread-safe against its own worker threads; the calls themselves are for one host thread. */
typedef struct mdbn_feeder mdbn_feeder;
int  mdbn_feeder_create(mdbn_ctx *ctx, const float *table, int64_t n_rows, int64_t cols, int64_t ld,
                        int64_t max_rows, int slots, float *const *device_slots, int64_t ld_device,
                        int threads, mdbn_feeder **out);
int  mdbn_feeder_submit(mdbn_feeder *f, const int64_t *indexes, int64_t n, int64_t *ticket);
int  mdbn_feeder_acquire(mdbn_feeder *f, int64_t ticket, void *stream, int *slot);
int  mdbn_feeder_release(mdbn_feeder *f, int64_t ticket, void *stream);
/* host-side time per stage since the last call: out5 = { minibatches fed, mean gather us, mean copy-enqueue us,
 * acquires, mean host wait in acquire us }; resets the counters */
int  mdbn_feeder_stats(mdbn_feeder *f, double *out5);
int  mdbn_feeder_cancel(mdbn_feeder *f);
int  mdbn_feeder_destroy(mdbn_feeder *f);

/* propup + sample_h_given_v (src/rbm.py:187-213), also HiddenLayer.output (src/mlp.py:103-107):
 * pre = v W + hbias ; mean = sigmoid(pre) ; sample = (u < mean).
 * pre / mean / sample may each be NULL; mean is stored multiplied by mean_scale. */
int  mdbn_propup_sample(mdbn_ctx *ctx, void *stream, const float *v, int64_t B, int64_t ldv,
                        const float *W, int64_t V, int64_t H, int64_t ldh, const float *hbias,
                        float *pre, float *mean, float mean_scale, float *sample,
                        const mdbn_rng *rng, void *workspace, int64_t workspace_bytes);

/* propdown + sample_v_given_h: RBM src/rbm.py:215-240 (gauss=0: sigmoid + Bernoulli),
 * GRBM src/rbm.py:647-660 (gauss=1: linear mean, sample = mean [+ N(0,1) if add_noise]).
 * If v0 != NULL the un-normalised reconstruction cost of src/rbm.py:479-480 / :697 is
 * written to cost_sum[0] (sum over all elements; caller divides). */
int  mdbn_propdown_sample(mdbn_ctx *ctx, void *stream, const float *h, int64_t B, int64_t ldh,
                          const float *W, int64_t V, int64_t H, int64_t ldv, const float *vbias,
                          int gauss, int add_noise, float *pre, float *mean, float *sample,
                          const mdbn_rng *rng, const float *v0, float *cost_sum,
                          void *workspace, int64_t workspace_bytes);

/* Data parallelism (added by this engine; the reference is single-process): one process per GPU, minibatch rows
 * sharded over the ranks, ONE sum all-reduce of the packed statistics per CD step over RCCL / xGMI.
 * mdbn_comm_unique_id fills 128 bytes on one rank (ncclGetUniqueId); the caller hands them to every rank (any
 * channel: torch.distributed's store, a file) and each calls mdbn_comm_init_rank (ncclCommInitRank) on its
 * context.  mdbn_allreduce_stats enqueues the in-place float32 sum of stats[0..n) on `stream`; like every entry
 * point it does not synchronise.  librccl.so is opened on first use. */
int  mdbn_comm_unique_id(char *id128);
int  mdbn_comm_init_rank(mdbn_ctx *ctx, const char *id128, int nranks, int rank);
int  mdbn_allreduce_stats(mdbn_ctx *ctx, void *stream, float *stats, int64_t n);
int  mdbn_comm_destroy(mdbn_ctx *ctx);

/* n_steps of gibbs_vhv (src/rbm.py:250-256; GRBM :673-682) in one call, in place and without allocation: the
 * sampling loop of src/rbm.py:806-865 (theano.scan over gibbs_vhv, 500 steps x 20 chains).  v [B, ldv] holds
 * the chain's visible state on entry and vis_samples[-1] on return; h_mean / h_sample [B, ldh] and v_mean
 * [B, ldv] are scratch during the chain and the LAST step's values on return (pre_h / pre_v: optional).
 * Random draws: step t (0-based) uses rng->step + 2t for the hidden and rng->step + 2t + 1 for the visible
 * draw, both with draw index 0 -- exactly what 2 * n_steps eager sample_* calls consume, so a chain equals
 * n_steps eager gibbs_vhv calls bit for bit; the caller advances its step counter by 2 * n_steps. */
int  mdbn_gibbs_chain(mdbn_ctx *ctx, void *stream, float *v, int64_t B, int64_t ldv, const float *W,
                      int64_t V, int64_t H, int64_t ldh, const float *hbias, const float *vbias, int gauss,
                      int add_noise, int64_t n_steps, float *pre_h, float *h_mean, float *h_sample,
                      float *pre_v, float *v_mean, const mdbn_rng *rng, void *workspace,
                      int64_t workspace_bytes);

/* compute_rbm_grad's products (src/rbm.py:411-417) from the stacked buffers of mdbn_cd_args:
 * stats = [S = v0'ph - nv'nh | s_h | s_v | (cost untouched)] */
int  mdbn_cd_stats(mdbn_ctx *ctx, void *stream, const float *V2, const float *P2, int64_t B,
                   int64_t V, int64_t H, int64_t ldv, int64_t ldh, float *stats,
                   void *workspace, int64_t workspace_bytes);

/* update rule of src/rbm.py:347-365 */
int  mdbn_apply_update(mdbn_ctx *ctx, void *stream, const mdbn_update_args *a);

/* gather + positive phase + k x gibbs_hvh + statistics (src/rbm.py:303-345, 374);
 * leaves the packed statistics in a->stats; follow with mdbn_apply_update (after the
 * all-reduce in data-parallel runs). */
int  mdbn_cd_step(mdbn_ctx *ctx, void *stream, const mdbn_cd_args *a);

/* mdbn_cd_step in two calls, for the overlapped data-parallel order: mdbn_cd_forward = gather + positive phase + Gibbs
 * chain (everything that reads the parameters), mdbn_cd_statistics = bias statistics + the statistics GEMM (reads only the
 * step's own buffers).  In between the caller waits for the all-reduce of the PREVIOUS step and hands its deferred update
 * (`deferred`: mdbn_update_args with phase 3 and the reduced statistics of step t-1; NULL = none) to the statistics call:
 * nothing of that update depends on this step's statistics GEMM, so on the plane path its weight half is applied by that
 * GEMM's loader waves during the main loop and its bias / cost half by the MFMA waves ahead of it -- no update launch;
 * otherwise (f32-operand kernels, balanced launches, lambda_1 != 0, fewer than 20 stages) the library launches the update
 * kernel itself right before the GEMM.  Bitwise the same parameters as mdbn_cd_step + mdbn_apply_update(phase 3).
 * mdbn_cd_statistics must directly follow mdbn_cd_forward with the same arguments on the same context. */
int  mdbn_cd_forward(mdbn_ctx *ctx, void *stream, const mdbn_cd_args *a);
int  mdbn_cd_statistics(mdbn_ctx *ctx, void *stream, const mdbn_cd_args *a, const mdbn_update_args *deferred);

/* The whole compiled step function for a single device (src/rbm.py:258-376): mdbn_cd_step
 * followed by the update, in one call; parameters, speeds and cost are bitwise identical to
 * mdbn_cd_step + mdbn_apply_update.  upd->phase is ignored.  With option "fused_update" (default)
 * the S block of a->stats is left unspecified; s_h, s_v and cost_sum are always written.  (Option "update_overlap" moves the bias / cost
 * finalisation and the parameter half of the update onto an internal side stream under the
 * statistics GEMM, which never reads W.) */
int  mdbn_cd_train_step(mdbn_ctx *ctx, void *stream, const mdbn_cd_args *a,
                        const mdbn_update_args *upd);

/* free energy: RBM src/rbm.py:166-171, GRBM src/rbm.py:684-688;  out[N] */
int  mdbn_free_energy(mdbn_ctx *ctx, void *stream, const float *x, int64_t N, int64_t ldv,
                      const float *W, int64_t V, int64_t H, int64_t ldh, const float *hbias,
                      const float *vbias, int gauss, float *out,
                      void *workspace, int64_t workspace_bytes);

/* Pieces of get_pseudo_likelihood_cost (src/rbm.py:421-447): out = round(x) (tensor.round: half away from
 * zero) with column flip_col replaced by 1 - round(x) (flip_col < 0: no flip); then, from the free energies of
 * the two matrices, cost_out[0] = -mean(n_visible * softplus(fe - fe_flip)). */
int  mdbn_round_flip(mdbn_ctx *ctx, void *stream, const float *x, int64_t rows, int64_t cols, int64_t ld,
                     int64_t flip_col, float *out);
int  mdbn_pl_cost(mdbn_ctx *ctx, void *stream, const float *fe, const float *fe_flip, int64_t rows,
                  int64_t n_visible, float *cost_out);
/* get_reconstruction_cost on given arrays: RBM src/rbm.py:449-482 (cross-entropy of sigmoid(pre) against
 * target, summed over units, mean over rows), GRBM src/rbm.py:690-699 (mean of (sigmoid(pre) - target)^2). */
int  mdbn_recon_cost(mdbn_ctx *ctx, void *stream, const float *pre, int64_t ld_pre, const float *target,
                     int64_t ld_target, int64_t rows, int64_t cols, int gauss, float *cost_out,
                     void *workspace, int64_t workspace_bytes);
/* HiddenLayer's tanh activation (src/mlp.py:36-110 default), in place on the live columns */
int  mdbn_tanh(mdbn_ctx *ctx, void *stream, float *x, int64_t rows, int64_t cols, int64_t ld);
/* count[0] += number of NaN / Inf values in x[0..n): the check NanGuardMode would make (src/rbm.py:542-543,
 * src/dbn.py:311); the caller zeroes count */
int  mdbn_count_nonfinite(mdbn_ctx *ctx, void *stream, const float *x, int64_t n, int32_t *count);

/* bfloat16 wire format of the data-parallel statistics (opt-in reporting mode MDBN_WIRE_BF16=1, never a
 * parity path): dst[i] = bfloat16(src[i]) with round-to-nearest-even, and the exact widening back.  n elements; both
 * buffers 16-byte aligned. */
int  mdbn_f32_to_bf16(mdbn_ctx *ctx, void *stream, const float *src, void *dst, int64_t n);
int  mdbn_bf16_to_f32(mdbn_ctx *ctx, void *stream, const void *src, float *dst, int64_t n);

/* the random matrices themselves (tests, and sampling utilities) */
int  mdbn_rng_uniform(mdbn_ctx *ctx, void *stream, float *out, int64_t rows, int64_t cols,
                      int64_t ld, const mdbn_rng *rng);
int  mdbn_rng_normal(mdbn_ctx *ctx, void *stream, float *out, int64_t rows, int64_t cols,
                     int64_t ld, const mdbn_rng *rng);
/* host-side twin of mdbn_rng_uniform (no GPU needed) */
int  mdbn_philox_host(float *out, int64_t rows, int64_t cols, int64_t ld, const mdbn_rng *rng);

#ifdef __cplusplus
}
#endif
#endif /* MDBN_HIP_H */
