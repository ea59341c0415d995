// C-ABI of libmdbn_hip.so (declared in include/mdbn_hip.h): argument validation, split-K
// planning, workspace carving and the launch sequence of one CD-k step.  No allocation,
// no synchronisation: every entry point only enqueues kernels on the caller's stream.
#include <hip/hip_runtime.h>
#include "row_pool.h"
#include <dlfcn.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <map>
#include <mutex>
#include <set>
#include <thread>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <utility>
#include <vector>
#include "mdbn_kernels.h"
#include "mdbn_small.h"
#include "mdbn_thin.h"
#include "mdbn_gchain.h"

using namespace mdbn;

// Tuning knobs of ONE context (mdbn_set_option(ctx, ...) writes ctx->opt; a knob set on one context never changes what
// another context of the process launches).  The option comments sit beside the g_opt_* names below, which read the
// options of the context whose call is running on this thread (CtxScope).
struct Options {
    int gemm_bk = 0;
    int x6_min_jobs = 48;
    int x6_pw = 4;
    int gemm_cw = 0;
    int update_overlap = 0;
    int fused_epilogue = 1;
    int fused_update = 1;
    int fused_finalize = 1;
    int skinny_gemm = 1;
    int skinny_fused_max_k = 1024;
    int64_t skinny_max_macs = 32ll << 20;
    int stream_x6 = 2;
    int64_t stream_max_macs = (int64_t)1 << 30;
    int stream_mi = 0;
    int stream_ni = 0;
    int gemm_bf16x6 = 3;
    int small_fused = 1;
    int thin_fused = 1;
    int gchain = 0;
    int gemm_planes = 1;
    int planes_mfma = 16;
    int early_w = 1;
    int narrow_tiles = 1;
    int feed_copy_streams = 1;
    int gather_ahead = 1;
    int bf16_inputs = 0;
    int64_t planes_min_work = (int64_t)1 << 30;
    int comm_cus = 0;
    int bal_blocks = 0;
    int min_splitk = 128;
    int epilogue_cw = 0;
    int epilogue_threads = 0;
    int small_fin_lanes = 0;
};

// Optional HIP-event timing of the GEMM launches (the dominant kernel), used by bench.py to
// quote the roofline from the kernel's own average duration on the stream it runs on.
struct GemmTiming {
    bool enabled = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pool;
    // per recorded launch: which GEMM it was and the work it issued (mdbn_kernel_timing_detail)
    struct Meta { int kind; double alg_flop, pipe_flop; };
    std::vector<Meta> meta;
    size_t used = 0;
};

struct mdbn_ctx {
    int device;
    int num_cu;
    Options opt;
    GemmTiming timing;                  // mdbn_kernel_timing: per context
    // mdbn_cd_forward -> mdbn_cd_statistics hand-over (host side only): the cost partials the chain's last visible pass left
    int pending_n_cost = -1;
    const void* pending_stats = nullptr;
    void* comm = nullptr;      // ncclComm_t of mdbn_comm_init_rank (RCCL), or NULL
    int comm_ranks = 0;
    // side stream + events for mdbn_cd_train_step (memory-bound update work overlapped with the
    // compute-bound statistics GEMM); created on first use
    hipStream_t side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    // group-chain step (mdbn_gchain.hip): the flag words of its in-launch exchange ([GC_MAX_FLAGS] + one error word; made
    // and zeroed on first use, written by that protocol only) and the sequence number of the next launch's first exchange
    unsigned* gc_flags = nullptr;
    unsigned gc_seq = 1;
};


// The options / timing records a library call reads are those of the context it was made on: every entry point that takes
// a context opens a CtxScope; context-free entry points (mdbn_workspace_bytes, mdbn_planes_eligible, ...) see the defaults
// of a fresh context.
static const Options g_default_options;
static GemmTiming g_no_timing;                          // calls outside any context record nothing
static thread_local const Options* t_opt = &g_default_options;
static thread_local GemmTiming* t_timing = &g_no_timing;
#define g_timing (*t_timing)
struct CtxScope {
    const Options* prev_opt; GemmTiming* prev_timing;
    explicit CtxScope(mdbn_ctx* c) : prev_opt(t_opt), prev_timing(t_timing)
    {
        if (c) {
            t_opt = &c->opt; t_timing = &c->timing;
            set_epilogue_cw(c->opt.epilogue_cw); set_epilogue_threads(c->opt.epilogue_threads);
            g_small_fin_lanes = c->opt.small_fin_lanes;
        }
    }
    ~CtxScope() { t_opt = prev_opt; t_timing = prev_timing; }
    CtxScope(const CtxScope&) = delete;
    CtxScope& operator=(const CtxScope&) = delete;
};

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_OK(expr)                                                                    \
    do {                                                                                \
        hipError_t _e = (expr);                                                         \
        if (_e != hipSuccess)                                                           \
            return fail(MDBN_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                        __FILE__, __LINE__);                                            \
    } while (0)

#define REQUIRE(cond, ...)                                 \
    do {                                                   \
        if (!(cond)) return fail(MDBN_EINVAL, __VA_ARGS__); \
    } while (0)

#define g_opt_gemm_bk (t_opt->gemm_bk)          // mdbn_set_option("gemm_bk"): 0 = auto, 32, 64
#define g_opt_x6_min_jobs (t_opt->x6_min_jobs)     // mdbn_set_option("x6_min_jobs"): fewer 128x128-tile jobs than this keep the exact kernel (192 -> 48: 1024->256 and 512^3 at B = 512 -6%)
#define g_opt_x6_pw (t_opt->x6_pw)            // mdbn_set_option("x6_producer_waves"): bf16x6 producer waves per operand (2 | 4); 4: step 162.4 -> 158.3 us
#define g_opt_gemm_cw (t_opt->gemm_cw)          // mdbn_set_option("gemm_cw"): MFMA waves per SIMD of the tiled GEMM, 0 = auto, 1, 2
// mdbn_set_option("update_overlap"): run finalize + the parameter half of the update on a side
// stream under the statistics GEMM.  Measured (profile r01j): the fork/join events cost more than
// the ~12 us they hide (278.7 vs 259.6 us per step), so it is off by default.
#define g_opt_update_overlap (t_opt->update_overlap)
// mdbn_set_option("fused_epilogue"): apply the activation epilogue on the MFMA accumulators when a
// GEMM needs no split-K (default on)
#define g_opt_fused_epilogue (t_opt->fused_epilogue)
// mdbn_set_option("fused_update"): mdbn_cd_train_step applies the update inside the statistics GEMM
// when that GEMM is not split, and lets the update kernel sum the split-K slabs when it is
// (default on); the S block of `stats` is then not materialised
#define g_opt_fused_update (t_opt->fused_update)
// mdbn_set_option("fused_finalize"): ... and runs the bias statistics / cost / bias update inside that GEMM too
#define g_opt_fused_finalize (t_opt->fused_finalize)
constexpr int kTargetJobs = 256;       // one 8-wave tile job per CU (MI355X: 256 CUs)
#define kMinSplitK (t_opt->min_splitk)     // >= 4 slices of BK = 32 per split (mdbn_set_option "gemm_min_splitk")

inline int64_t ru4(int64_t x) { return (x + 3) & ~int64_t(3); }
// Leading-dimension policy (mdbn_padded_ld).  Padding 1-KiB-multiple rows by 64 floats (to spread
// a GEMM slice's rows over the L2 channels) gained ~10% in the isolated GEMM microbenchmark
// (scripts/gemm_ldpad.py) but LOST 2% on the whole CD step (K1 42.3 vs 40.6 us, update 17.0 vs
// 13.5 us, profiles r01e vs r01d), and does nothing for the streaming kernel of the mid-size layers either
// (profiles/r05zc_ldpad.log), so the policy is plain round_up(cols, 4); every entry point
// accepts any ld % 4 == 0, ld >= cols.
inline int64_t padded_ld(int64_t cols) { return ru4(cols); }
inline int64_t ru64(int64_t x) { return (x + 63) & ~int64_t(63); }
inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

struct Plan {
    int tiles_m, tiles_n, splitk, kchunk, bn, bk;
    int skinny = 0;        // skinny_gemm_kernel: tiles_m x tiles_n = (32*mi)-row tiles x 32-column strips
    int mi = 1;
    int ni = 1;            // streaming bf16x6 kernel: strips per tile
    int cw = 1;            // MFMA (consumer) waves per SIMD of the tiled kernel
    int x6 = 0;            // statistics GEMM on the bf16 pipe (stats_bf16x6_kernel)
    int64_t slab_floats(int64_t M, int64_t ldc) const { return (int64_t)splitk * M * ldc; }
    void fill(GemmArgs& g) const
    {
        g.kchunk = kchunk; g.splitk = splitk; g.tiles_m = tiles_m; g.tiles_n = tiles_n; g.bn = bn; g.bk = bk;
        g.inner_m = tiles_m <= tiles_n;
        g.skinny = skinny; g.mi = mi; g.ni = ni; g.fused = 0; g.cw = cw; g.fin_enabled = 0; g.x6 = x6; g.x6_pw = g_opt_x6_pw;
    }
};

// Tile shape and split-K: 128x128 tiles whenever they alone give every CU a job; smaller
// problems keep the 128x128 tile too and split the reduction until about one job per CU
// exists (never below kMinSplitK of reduction per job); 128x64 tiles are used only when even
// that leaves CUs idle.
Plan plan_gemm(int64_t M, int64_t N, int64_t K)
{
    Plan p;
    p.tiles_m = (int)((M + 127) / 128);
    const int64_t t128 = (int64_t)p.tiles_m * ((N + 127) / 128);
    p.bn = (t128 * std::max<int64_t>(1, K / kMinSplitK) >= kTargetJobs) ? 128 : 64;
    p.tiles_n = (int)((N + p.bn - 1) / p.bn);
    const int64_t tiles = (int64_t)p.tiles_m * p.tiles_n;
    int64_t want = std::max<int64_t>(1, kTargetJobs / std::max<int64_t>(tiles, 1));
    int64_t maxsplit = std::max<int64_t>(1, K / kMinSplitK);
    int64_t sk = std::min(want, maxsplit);
    // slice depth: 64 halves the per-slice barrier / pipe-refill overhead; it needs the 128x128
    // tile (LDS) and at least two slices per job
    p.bk = (p.bn == 128 && g_opt_gemm_bk != 32 && (K + sk - 1) / sk >= 128) ? 64 : 32;
    int64_t kchunk = ((K + sk - 1) / sk + p.bk - 1) / p.bk * p.bk;
    if (kchunk < p.bk) kchunk = p.bk;
    sk = std::max<int64_t>(1, (K + kchunk - 1) / kchunk);
    p.splitk = (int)sk;
    p.kchunk = (int)kchunk;
    // Two MFMA waves per SIMD (eight 64x32 wave tiles) gain 1-3% on short jobs (<= 8 slices: faster
    // ramp and drain; c2 step 240.1 -> 234.0 us) and lose up to 10% in steady state (1.5 instead of
    // 1.0 LDS fragment dwords per MFMA; a second wave does NOT hide the LDS-return cost):
    // scripts/gemm_cw_ab.py.
    p.cw = g_opt_gemm_cw ? g_opt_gemm_cw : (p.bn == 128 && p.kchunk / p.bk <= 8 ? 2 : 1);
    return p;
}

// mdbn_set_option("skinny_gemm"): route GEMMs of <= 64 output rows, and small-layer GEMMs whose
// operands are L2-resident, to the register-streaming skinny_gemm_kernel (default on)
#define g_opt_skinny_gemm (t_opt->skinny_gemm)
constexpr int kSkinnyTargetBlocks = 512;   // two 8-wave blocks per CU
constexpr int kSkinnyMinK = 256;           // >= 4 octets per wave
#define g_opt_skinny_fused_max_k (t_opt->skinny_fused_max_k)   // up to here one block streams the whole K range
// Above 64 rows the streaming kernel only pays for problems so small that the chain of dependent
// launches is the whole cost; every register batch exposes an L2 latency, so long per-wave K
// streams (K > 512) get a third of the budget.  Measured at B = 512 (scripts/skinny_macs_ab.py):
// 400->40 76 -> 52 us, 256->200 72 -> 48, 100->128 69 -> 41, 512->40 CD-5 164 -> 124; 1024->256
// (134 M MACs per pass) 76 either way; 256x200 statistics over K = 1024 (52 M) 5 us slower.
#define g_opt_skinny_max_macs (t_opt->skinny_max_macs)

// Skinny plan of out[M, N (ldo stored)] over K: (32*mi)-row tiles x 32-column strips x K ranges.
Plan plan_skinny(int64_t M, int64_t K, int64_t ldo, bool allow_split)
{
    Plan p;
    p.skinny = 1;
    p.mi = M <= 32 ? 1 : 2;
    p.tiles_m = (int)((M + 32 * p.mi - 1) / (32 * p.mi));
    p.tiles_n = (int)((ldo + 31) / 32);
    p.bn = 32; p.bk = 8;
    const int64_t blocks = (int64_t)p.tiles_m * p.tiles_n;
    const int64_t want = std::max<int64_t>(1, kSkinnyTargetBlocks / blocks);
    // a K range per block only pays once the per-wave stream (K / 8) is long: an extra epilogue
    // launch costs ~5 us, 1024 of K cost a wave ~2 us of MFMA issue
    const bool one_launch = !allow_split || (K <= g_opt_skinny_fused_max_k && K <= 64 * blocks);
    const int64_t sk = one_launch ? 1 : std::min(want, std::max<int64_t>(1, K / kSkinnyMinK));
    int64_t kchunk = ((K + sk - 1) / sk + 63) / 64 * 64;
    p.kchunk = (int)kchunk;
    p.splitk = (int)std::max<int64_t>(1, (K + kchunk - 1) / kchunk);
    return p;
}

// Does the register-streaming kernel serve out[M, N] over K better than the LDS-tiled one?  Yes for
// <= 64 rows (no operand reuse to stage for), and for tiny problems, where one launch with a fused
// epilogue beats GEMM + slabs + epilogue kernel and every operand re-read stays in L2.
bool prefer_skinny(int64_t M, int64_t N, int64_t K)
{
    if (!g_opt_skinny_gemm) return false;
    if (M <= 64) return true;
    if (K > g_opt_skinny_fused_max_k) return false;
    const int64_t tiles_m = (M + 63) / 64, strips = (N + 31) / 32;
    if (K > 64 * tiles_m * strips) return false;
    return M * N * K <= (K <= 512 ? 3 : 1) * g_opt_skinny_max_macs;
}


// mdbn_set_option("gemm_bf16x6"): GEMMs of full 128x128 tiles run on the bf16 matrix pipe with three-way
// split operands (f32 accuracy, see gemm_bf16x6_kernel); bit 0 = statistics GEMM, bit 1 = forward
// passes; default 3
#define g_opt_gemm_bf16x6 (t_opt->gemm_bf16x6)
// mdbn_set_option("gemm_planes"): the CD step runs on pre-split bf16 planes (mdbn_planes.hip) when the caller
// supplies the plane buffers and the shape is made of whole 128-row / 128-column tiles (default on)
// mdbn_set_option("small_fused"): layers whose W fits one CU's LDS run the whole CD-k chain in ONE launch + a small finish
// launch (mdbn_small.hip); 0 = the multi-launch path (register-streaming GEMMs, fused epilogues)
#define g_opt_small_fused (t_opt->small_fused)
// mdbn_set_option("thin_fused"): minibatches of <= 32 rows (the reference's batch_size = 20) on layers that are not
// LDS-resident run the stream-over-W step of mdbn_thin.hip (W read 2 + k times per CD-k step); 0 = the register-streaming
// GEMM path
#define g_opt_thin_fused (t_opt->thin_fused)
// mdbn_set_option("gchain") (default 0): mid-size layers at B > 32 whose W fits the LDS of a GROUP of 2-16 CUs (256 -> 200,
// 1024 -> 256) run the positive phase and the whole Gibbs chain in ONE launch (mdbn_gchain.hip).  Parity-green and
// deterministic, but NOT faster than one launch per pass (profiles/r05q_gchain_ab.log: 119.8 vs 104.9 us at 256 -> 200
// CD-5, 68.6 vs 65.8 at 1024 -> 256): an in-launch exchange + the epilogues behind it cost what a kernel boundary costs.
#define g_opt_gchain (t_opt->gchain)
#define g_opt_gemm_planes (t_opt->gemm_planes)
// mdbn_set_option("planes_mfma"): MFMA shape of the plane GEMMs: 16 = v_mfma_f32_16x16x32_bf16 (default: the chip holds
// a higher clock on it), 32 = v_mfma_f32_32x32x16_bf16 (the products and order of gemm_bf16x6_kernel: same bits as the
// f32-operand path)
#define g_opt_planes_mfma (t_opt->planes_mfma)
// mdbn_set_option("early_w"): the statistics GEMM's loader waves apply the parameter half of the fused update during the
// main loop (W' needs only the old W and the old speed), the epilogue only forms the new speed (default on; same bits)
#define g_opt_early_w (t_opt->early_w)
// mdbn_set_option("narrow_tiles"): forward passes whose 128 x 128 plan would split K two ways run unsplit on 128 x 64 tiles
// with the fused activation epilogue instead (default on)
#define g_opt_narrow_tiles (t_opt->narrow_tiles)
// mdbn_set_option("feed_copy_streams"): a row feeder created afterwards moves each minibatch as 1 or 2 copies on as many
// streams (two SDMA engines side by side)
#define g_opt_feed_copy_streams (t_opt->feed_copy_streams)
// mdbn_set_option("gather_ahead"): honour mdbn_cd_args.next_indexes (default on; same bits)
#define g_opt_gather_ahead (t_opt->gather_ahead)
// mdbn_set_option("bf16_inputs"): REPORTING mode of BASELINE configs[1] ("bf16/fp32"): the plane GEMMs use only the
// leading bf16 piece of every operand (inputs truncated to bf16, f32 accumulation, one product instead of six).
// Probabilities then carry ~4e-3 of error: never used for a parity claim, off by default.
#define g_opt_bf16_inputs (t_opt->bf16_inputs)
// mdbn_set_option("planes_min_work"): the plane path serves a whole-tile shape only from B * V * H >= this on (and V * H
// >= 2^21): below, the launches of a step are so short that writing every tensor twice (f32 + planes) costs more than the
// cheaper GEMMs gain (c4's second layer 1024 -> 256 at B = 512: 69.2 us on the f32-operand kernels, 77.3 on planes;
// 2048 -> 1024: 122.9 vs 118.9; c2: 158.7 vs 150).  0: every whole-tile shape (tests).
#define g_opt_planes_min_work (t_opt->planes_min_work)
// mdbn_set_option("comm_cus"): CUs left to a collective that runs beside the step (data-parallel mode).  > 0: the plane
// GEMMs of mdbn_cd_step are launched BALANCED on (CUs - comm_cus) workgroups (mdbn_planes.hip, "BALANCED launches"):
// a collective's kernel takes whole CUs, and a one-workgroup-per-CU grid on fewer CUs needs a second round (measured:
// 163 -> 219 us per step with 8 CUs taken, scripts/dp_contention_probe.py).  0 (default): one workgroup per CU.
#define g_opt_comm_cus (t_opt->comm_cus)
#define g_opt_bal_blocks (t_opt->bal_blocks)       // "bal_blocks" (tests): the number of workgroups itself, whatever the device has
constexpr int kMaxBalBlocks = 256;

// Turn an LDS-tiled plan into a bf16x6 plan (128x128 tiles, 32-deep slices) when that leaves enough
// jobs to spread over the chip; `unsplit` = the caller needs splitk == 1 (fused statistics epilogue).
bool try_bf16x6(Plan& p, int64_t M, int64_t N, int64_t K, bool unsplit = false)
{
    if (p.skinny || K < 64) return false;
    const int64_t tm = (M + 127) / 128, tn = (N + 127) / 128, tiles = tm * tn;
    int64_t sk = 1;
    if (!unsplit) {
        sk = std::min(std::max<int64_t>(1, kTargetJobs / tiles), std::max<int64_t>(1, K / kMinSplitK));
    } else if (p.splitk != 1) {
        return false;
    }
    if (tiles * sk < g_opt_x6_min_jobs) return false;  // too few jobs: the exact kernel's 128x64 tiles spread wider
    int64_t kchunk = ((K + sk - 1) / sk + 31) / 32 * 32;
    sk = (K + kchunk - 1) / kchunk;
    p.x6 = 1;
    p.bn = 128; p.bk = 32; p.cw = 1;
    p.tiles_m = (int)tm; p.tiles_n = (int)tn;
    p.splitk = (int)sk; p.kchunk = (int)kchunk;
    return true;
}

// mdbn_set_option("stream_x6") (default 2): mid-size passes at more than 64 rows -- too small for 128 x 128 tiles without a
// split-K + slab + epilogue-launch round trip -- run UNSPLIT on 32 x 32 (64 x 32) tiles of the register-streaming kernel on
// the bf16 matrix pipe (mdbn_stream.hip: f32 operands split in registers, six / three piece products, fused epilogue): one
// launch per pass.  2: the small-layer passes prefer_skinny() sends to the exact-f32 streaming kernel use it as well
// (256 -> 200 at B = 512: CD-1 37.9 -> 30.1 us, CD-5 100.1 -> 71.6: profiles/r05za_configs_ab.log); 1: only the passes
// the LDS-tiled kernels served.
// "stream_max_macs": largest M * N * K served; "stream_mi": 0 = auto, 1 | 2 = 32-row blocks per tile.
#define g_opt_stream_x6 (t_opt->stream_x6)
#define g_opt_stream_max_macs (t_opt->stream_max_macs)
#define g_opt_stream_mi (t_opt->stream_mi)
#define g_opt_stream_ni (t_opt->stream_ni)
constexpr int64_t kStreamMinTiles = 192;

bool stream_ok(int64_t M, int64_t N, int64_t K)
{
    return g_opt_stream_x6 && M > 64 && K >= 64 && M * N * K <= g_opt_stream_max_macs;
}

Plan plan_stream(int64_t M, int64_t K, int64_t ldo)
{
    Plan p = plan_skinny(M, K, ldo, false);          // one K range: the epilogue runs on the tile
    // the largest tile that still leaves about one workgroup per CU: a 64 x 64 tile moves half the operand bytes of four
    // 32 x 32 ones through L2 and splits every fragment once for two products
    const int64_t tiles32 = ((M + 31) / 32) * ((ldo + 31) / 32);
    p.mi = tiles32 >= 2 * kStreamMinTiles ? 2 : 1;
    p.ni = tiles32 >= 4 * kStreamMinTiles ? 2 : 1;
    if (g_opt_stream_mi) p.mi = g_opt_stream_mi;
    if (g_opt_stream_ni) p.ni = g_opt_stream_ni;
    p.tiles_m = (int)((M + 32 * p.mi - 1) / (32 * p.mi));
    p.x6 = 1;
    return p;
}

// Plan of one forward pass (x[M, K] * op(W) -> [M, N], `ldo` columns stored).
// stream = false: the plan of the LDS-tiled kernels whatever the streaming option says (the plane path runs on their tiling)
Plan plan_forward(int64_t M, int64_t N, int64_t K, int64_t ldo, bool stream = true)
{
    if (prefer_skinny(M, N, K)) {
        if (stream && g_opt_stream_x6 >= 2 && stream_ok(M, N, K)) return plan_stream(M, K, ldo);
        return plan_skinny(M, K, ldo, M <= 64);
    }
    if (stream && stream_ok(M, N, K)) return plan_stream(M, K, ldo);
    Plan p = plan_gemm(M, N, K);
    if (g_opt_gemm_bf16x6 & 2) try_bf16x6(p, M, N, K);
    return p;
}

// Plan of the statistics GEMM S[V, H] = V2^T P2 over K = 2B.
Plan plan_stats(int64_t V, int64_t H, int64_t K2, int64_t ldh, bool stream = true)
{
    if (prefer_skinny(V, H, K2) && V > 64) {
        if (stream && g_opt_stream_x6 >= 2 && stream_ok(V, H, K2)) return plan_stream(V, K2, ldh);
        return plan_skinny(V, K2, ldh, false);
    }
    if (stream && stream_ok(V, H, K2)) return plan_stream(V, K2, ldh);
    Plan p = plan_gemm(V, H, K2);
    if (g_opt_gemm_bf16x6 & 1) try_bf16x6(p, V, H, K2, p.splitk == 1);     // unsplit stays unsplit (fused epilogues)
    return p;
}

static unsigned long long* g_stamps = nullptr;     // diagnostic builds only

hipError_t timed_gemm(int la, int lb, const GemmArgs& g_in, hipStream_t s)
{
    GemmArgs g = g_in;
    g.stamps = g_stamps;
    if (!g_timing.enabled || g_timing.used >= 8192) return launch_gemm(la, lb, g, s);
    if (g_timing.used == g_timing.pool.size()) {
        hipEvent_t a, b;
        hipError_t e = hipEventCreate(&a);
        if (e != hipSuccess) return e;
        e = hipEventCreate(&b);
        if (e != hipSuccess) return e;
        g_timing.pool.emplace_back(a, b);
    }
    {
        // kind = 100 * pipe (0 exact-f32 MFMA, 1 bf16 pipe with 6 products, 2 bf16 pipe with 3) + 10 * fused
        //        + 2 * la + lb; (la, lb) = (K, MN) propup, (K, K) propdown, (MN, MN) statistics
        const double alg = 2.0 * (double)g.M * (double)g.N * (double)g.K;
        const int pipe = g.x6;
        GemmTiming::Meta m{100 * pipe + 10 * g.fused + 2 * la + lb + (g.skinny ? 1000 : 0), alg,
                           alg * (pipe == 1 ? 6.0 : pipe == 2 ? 3.0 : 1.0)};
        if (g_timing.meta.size() <= g_timing.used) g_timing.meta.resize(g_timing.used + 1);
        g_timing.meta[g_timing.used] = m;
    }
    auto& ev = g_timing.pool[g_timing.used++];
    hipError_t e = hipEventRecord(ev.first, s);
    if (e != hipSuccess) return e;
    e = launch_gemm(la, lb, g, s);
    if (e != hipSuccess) return e;
    return hipEventRecord(ev.second, s);
}

PhiloxKey make_key(const mdbn_rng& r, uint32_t draw)
{
    PhiloxKey k;
    k.k0 = (uint32_t)r.seed;
    k.k1 = (uint32_t)(r.seed >> 32) ^ r.stream_id;
    k.step = r.step;
    k.draw = draw;
    k.row_offset = r.row_offset;
    return k;
}

// carve-up of the caller's workspace
struct Workspace {
    float* slabs;
    int64_t slab_floats;
    float* cost_partials;
    int64_t cost_floats;
    float* colPpos;   // [row_groups][ldh]  4-row partial column sums of  ph_mean
    float* colPneg;   // [row_groups][ldh]                                -nh_mean
    float* colV;      // [row_groups][ldv]                                 v0 - nv_mean
};

struct WsSizes {
    int64_t slab, cost, colP, colV;
    int64_t total_bytes() const { return 4 * (ru64(slab) + ru64(cost) + ru64(colP) + ru64(colV)); }
};

static WsSizes ws_sizes_dense(int64_t B, int64_t V, int64_t H)
{
    const int64_t ldv = padded_ld(V), ldh = padded_ld(H);      // the largest ld the policy hands out
    WsSizes s;
    const Plan up = plan_gemm(B, H, V), down = plan_gemm(B, V, H), st = plan_gemm(V, H, 2 * B);
    s.slab = std::max(up.slab_floats(B, ldh), down.slab_floats(B, ldv));
    {       // whichever kernel serves the pass under the current options (skinny / bf16x6 / exact tiles)
        const Plan ups = plan_forward(B, H, V, ldh), downs = plan_forward(B, V, H, ldv);
        s.slab = std::max(s.slab, std::max(ups.slab_floats(B, ldh), downs.slab_floats(B, ldv)));
    }
    if (st.splitk > 1) s.slab = std::max(s.slab, st.slab_floats(V, ldh));
    {
        const Plan st2 = plan_stats(V, H, 2 * B, ldh);        // under the current options (bf16x6 splits differently)
        if (st2.splitk > 1) s.slab = std::max(s.slab, st2.slab_floats(V, ldh));
    }
    if (B % 128 == 0 && V % 128 == 0 && H % 128 == 0) {
        // balanced launches (comm_cus > 0): a tile has at most P / tiles + 2 segments, P <= kMaxBalBlocks; the in-place
        // statistics GEMM parks 64 KB per workgroup
        const int64_t tu = (B / 128) * (H / 128), td = (B / 128) * (V / 128);
        s.slab = std::max(s.slab, (kMaxBalBlocks / tu + 2) * B * ldh);
        s.slab = std::max(s.slab, (kMaxBalBlocks / td + 2) * B * ldv);
        s.slab = std::max<int64_t>(s.slab, (int64_t)kMaxBalBlocks * 2 * 16384);
    }
    if (small_shape_ok(B, V, H, 0) || small_shape_ok(B, V, H, 1))       // one S partial per workgroup of the one-launch step
        s.slab = std::max<int64_t>(s.slab, (int64_t)small_blocks(B) * ((V + 63) & ~int64_t(63)) * ldh);   // (pad lanes included)
    int64_t thin_cost = 0;
    {       // thin-batch step (mdbn_thin.hip): one [Bq, ldh] partial and one cost partial per workgroup of a pass
        ThinGeom tg;
        if (thin_geom(B, V, H, ldv, ldh, kTargetJobs, tg)) {
            s.slab = std::max<int64_t>(s.slab, (int64_t)tg.G * tg.Bq * ldh);
            thin_cost = tg.G;
        }
    }
    int64_t gc_cost = 0;
    {       // group-chain step (mdbn_gchain.hip): the exchange payload lives in the slab region (the statistics GEMM's slabs
            // come after the chain), one cost partial per (slab, member)
        GChainGeom gg;
        if (gchain_geom(B, V, H, ldv, ldh, 0, kTargetJobs, gg)) {
            s.slab = std::max<int64_t>(s.slab, gg.xbuf_floats);
            gc_cost = (int64_t)gg.nslab * gg.g;
        }
    }
    s.slab = std::max<int64_t>(s.slab, 4 * std::max(ldv, ldh) * 8);
    s.cost = std::max<int64_t>(256, (((B + 3) / 4) * std::max(ldv, ldh) + 63) / 64) + 64;   // worst case: one column per thread, 64-thread blocks
    // fused epilogues write one partial per block: 128 x 64 tiles, or 32-column strips (skinny)
    s.cost = std::max<int64_t>(s.cost, ((B + 127) / 128) * ((std::max(ldv, ldh) + 63) / 64) + 64);
    s.cost = std::max<int64_t>(s.cost, ((B + 31) / 32) * ((std::max(ldv, ldh) + 31) / 32) + 64);
    if (small_shape_ok(B, V, H, 0) || small_shape_ok(B, V, H, 1)) s.cost = std::max<int64_t>(s.cost, (int64_t)small_blocks(B) * SM_NW + 64);   // a cost partial per wave
    s.cost = std::max<int64_t>(s.cost, thin_cost + 64);
    s.cost = std::max<int64_t>(s.cost, gc_cost + 64);
    const int ng = row_groups(B);
    s.colP = 2 * (int64_t)ng * ldh;
    s.colV = (int64_t)ng * ldv;
    return s;
}

// (a ragged hidden width may ride on a leading dimension padded to a multiple of 128 -- plane_shape_ok -- : every buffer is
// then as large as the dense layer of that width needs)
WsSizes ws_sizes(int64_t B, int64_t V, int64_t H)
{
    WsSizes s = ws_sizes_dense(B, V, H);
    if (H % 128 != 0) {
        const WsSizes p = ws_sizes_dense(B, V, (H + 127) & ~int64_t(127));
        s.slab = std::max(s.slab, p.slab); s.cost = std::max(s.cost, p.cost);
        s.colP = std::max(s.colP, p.colP); s.colV = std::max(s.colV, p.colV);
    }
    return s;
}

int carve(void* ws, int64_t bytes, int64_t B, int64_t V, int64_t H, Workspace& out, bool need_stats)
{
    REQUIRE(ws != nullptr && aligned16(ws), "workspace must be a 16-byte aligned device pointer");
    const WsSizes s = ws_sizes(B, V, H);
    float* p = reinterpret_cast<float*>(ws);
    if (need_stats) {
        if (bytes < s.total_bytes())
            return fail(MDBN_ENOSPC, "workspace %lld bytes < %lld needed for B=%lld V=%lld H=%lld",
                        (long long)bytes, (long long)s.total_bytes(), (long long)B, (long long)V, (long long)H);
        out.slabs = p;            out.slab_floats = s.slab;  p += ru64(s.slab);
        out.cost_partials = p;    out.cost_floats = s.cost;  p += ru64(s.cost);
        out.colPpos = p;          out.colPneg = p + s.colP / 2;  p += ru64(s.colP);
        out.colV = p;
    } else {
        // propagation only: a fixed cost region at the end, everything else is slabs
        const int64_t floats = bytes / 4;
        const int64_t cost = 1 << 16;
        if (floats < cost + 4096)
            return fail(MDBN_ENOSPC, "workspace %lld bytes is too small", (long long)bytes);
        out.slabs = p;
        out.slab_floats = (floats - cost) & ~int64_t(63);
        out.cost_partials = p + out.slab_floats;
        out.cost_floats = cost;
        out.colPpos = out.colPneg = out.colV = nullptr;
    }
    return MDBN_OK;
}

// One affine map + activation over `rows` rows, chunked so the split-K slabs fit.
//   up   (dir 0): x[rows, V] * W        -> [rows, H]   (bias = hbias)
//   down (dir 1): x[rows, H] * W^T      -> [rows, V]   (bias = vbias)
struct Affine {
    const float* x; int64_t rows, ldx;
    const float* W; int64_t V, H, ldw;
    int dir;
    const float* bias;
    float* pre; float* mean; float* sample; int64_t ldo;
    float mean_scale; int gauss;
    const float* target; int64_t ld_target;
    bool want_cost;
    const mdbn_rng* rng; uint32_t draw;
    float* colsum = nullptr; int colsum_kind = 0;
    // x holds 0/1 samples written by our own epilogues (a Gibbs chain state): exactly representable
    // in bf16, so the bf16x6 kernel needs one piece of it and three products instead of six
    bool x_binary = false;
};

int run_affine(const Affine& a, const Workspace& ws, hipStream_t s, int* n_cost_out)
{
    const int64_t Kdim = a.dir == 0 ? a.V : a.H;
    const int64_t Ndim = a.dir == 0 ? a.H : a.V;
    int n_cost = 0;
    int64_t r0 = 0;
    while (r0 < a.rows) {
        int64_t R = a.rows - r0;
        Plan p = plan_forward(R, Ndim, Kdim, a.ldo);
        bool fuse = false;
        for (;;) {      // shrink the chunk until its slabs (unfused) and cost partials fit
            fuse = g_opt_fused_epilogue && p.splitk == 1;
            const bool slabs_fit = fuse || p.slab_floats(R, a.ldo) <= ws.slab_floats;
            const int64_t need_cost = !a.want_cost ? 0 : fuse ? (int64_t)p.tiles_m * p.tiles_n : epilogue_blocks(R, a.ldo);
            if (slabs_fit && n_cost + need_cost <= ws.cost_floats) break;
            if (R <= 4) return fail(MDBN_ENOSPC, "workspace cannot hold one 4-row chunk");
            R = std::max<int64_t>(4, (R / 2 + 3) & ~int64_t(3));
            p = plan_forward(R, Ndim, Kdim, a.ldo);
        }
        GemmArgs g{};
        g.A = a.x + r0 * a.ldx;  g.lda = a.ldx;
        g.B = a.W;               g.ldb = a.ldw;
        g.C = ws.slabs;          g.ldc = a.ldo;
        g.slab_stride = R * a.ldo;
        g.M = (int)R; g.N = (int)Ndim; g.K = (int)Kdim; g.Nst = (int)a.ldo;
        p.fill(g);
        if (p.x6 && a.x_binary) g.x6 = 2;

        EpiArgs e{};
        e.slabs = ws.slabs; e.slab_stride = g.slab_stride; e.nsplit = p.splitk;
        e.rows = (int)R; e.cols = (int)Ndim; e.ld = a.ldo;
        e.bias = a.bias;
        e.pre = a.pre ? a.pre + r0 * a.ldo : nullptr;
        e.mean = a.mean ? a.mean + r0 * a.ldo : nullptr;
        e.sample = a.sample ? a.sample + r0 * a.ldo : nullptr;
        e.mean_scale = a.mean_scale; e.gauss = a.gauss;
        e.target = a.target ? a.target + r0 * a.ld_target : nullptr;
        e.ld_target = a.ld_target;
        e.colsum = a.colsum ? a.colsum + (r0 / 4) * a.ldo : nullptr;
        e.colsum_kind = a.colsum_kind;
        const int nb = fuse ? p.tiles_m * p.tiles_n : epilogue_blocks(R, a.ldo);
        e.cost_partials = nullptr;
        if (a.want_cost) {
            if (n_cost + nb > ws.cost_floats) return fail(MDBN_ENOSPC, "cost scratch exhausted");
            e.cost_partials = ws.cost_partials + n_cost;
            n_cost += nb;
        }
        mdbn_rng zero;
        memset(&zero, 0, sizeof zero);
        const mdbn_rng& rr = a.rng ? *a.rng : zero;
        e.rng = make_key(rr, a.draw);
        e.rng.row_offset = rr.row_offset + (uint64_t)r0;
        if (fuse) {                 // activation on the accumulators: no slabs, one launch
            g.fused = 1;
            g.epi = e;
            HIP_OK(timed_gemm(LAY_K, a.dir == 0 ? LAY_MN : LAY_K, g, s));
        } else {
            HIP_OK(timed_gemm(LAY_K, a.dir == 0 ? LAY_MN : LAY_K, g, s));
            HIP_OK(launch_act_epilogue(e, s));
        }
        r0 += R;
    }
    if (n_cost_out) *n_cost_out = n_cost;
    return MDBN_OK;
}

int check_mat(const void* p, int64_t ld, int64_t cols, const char* name)
{
    REQUIRE(p != nullptr, "%s is NULL", name);
    REQUIRE(aligned16(p), "%s is not 16-byte aligned", name);
    REQUIRE(ld % 4 == 0 && ld >= cols, "%s: leading dimension %lld must be a multiple of 4 and >= %lld",
            name, (long long)ld, (long long)cols);
    return MDBN_OK;
}

#define CHECK(expr) do { int _rc = (expr); if (_rc != MDBN_OK) return _rc; } while (0)


// ---------------------------------------------------------------------------------- bf16 plane path
hipError_t timed_gemm_planes(int la, int lb, const PlaneGemmArgs& g_in, hipStream_t s)
{
    PlaneGemmArgs g = g_in;
    g.stamps = g_stamps;
    g.ms = g_opt_planes_mfma;
    if (g_opt_bf16_inputs) g.ap = 0;
    auto launch = [&]() { return g.bal ? launch_gemm_planes_bal(la, lb, g, s) : launch_gemm_planes(la, lb, g, s); };
    if (!g_timing.enabled || g_timing.used >= 8192) return launch();
    if (g_timing.used == g_timing.pool.size()) {
        hipEvent_t a, b;
        hipError_t e = hipEventCreate(&a);
        if (e != hipSuccess) return e;
        e = hipEventCreate(&b);
        if (e != hipSuccess) return e;
        g_timing.pool.emplace_back(a, b);
    }
    {
        const double alg = 2.0 * (double)g.M * (double)g.N * (double)g.K;
        const int pipe = g.ap == 3 ? 1 : (g.ap == 1 ? 2 : 3);           // 3: one product (bf16-input reporting mode)
        GemmTiming::Meta m{(g.bal ? 3000 : 2000) + 100 * pipe + 10 * g.fused + 2 * la + lb, alg, alg * (pipe == 1 ? 6.0 : pipe == 2 ? 3.0 : 1.0)};
        if (g_timing.meta.size() <= g_timing.used) g_timing.meta.resize(g_timing.used + 1);
        g_timing.meta[g_timing.used] = m;
    }
    auto& ev = g_timing.pool[g_timing.used++];
    hipError_t e = hipEventRecord(ev.first, s);
    if (e != hipSuccess) return e;
    e = launch();
    if (e != hipSuccess) return e;
    return hipEventRecord(ev.second, s);
}

// carve-up of mdbn_cd_args.planes (bf16 elements): planes of X2 = [v0; nv] and P2 = [ph; -nh], hs, vs
struct PlaneBufs {
    unsigned short *Xp, *Pp, *hsp, *vsp;
    int64_t px, pp;          // elements between the planes of Xp / Pp
};
inline int64_t planes_elems(int64_t B, int64_t ldv, int64_t ldh) { return 6 * B * ldv + 6 * B * ldh + B * ldh + B * ldv; }

// The plane GEMMs take over exactly the problems the bf16x6 plans cover (whole 128x128 tiles, 32-deep slices, the
// same split factors): a plane step and an f32-operand step then sum in the same order and agree bit for bit.
bool plane_plan(const Plan& p, int64_t M, int64_t N, int64_t K)
{
    return p.x6 && !p.skinny && M % 128 == 0 && N % 128 == 0 && p.kchunk % 32 == 0 && (int64_t)p.kchunk * p.splitk == K;
}

// the shape half of the rule (mdbn_planes_eligible: the host allocates plane buffers only for shapes this accepts)
bool plane_shape_ok(int64_t B, int64_t V, int64_t H, int64_t ldv, int64_t ldh)
{
    if (!g_opt_gemm_planes) return false;
    // A ragged hidden width rides on a padded leading dimension: the GEMMs run on He = ldh columns (a multiple of 128; the
    // pad columns of W, of the hidden activations and of the statistics hold exact zeros, which every kernel of the path
    // keeps so), the activation epilogues treat the columns >= H as dead.  The visible side stays dense: its leading
    // dimension is the training table's.
    if (B <= 0 || B % 128 || V % 128 || ldh % 128 || ldv != V || H > ldh || ldh - H >= 128 || B > 65535) return false;
    const int64_t He = ldh;
    if (g_opt_planes_min_work > 0 && (B * V * He < g_opt_planes_min_work || V * He < ((int64_t)1 << 21))) return false;
    return plane_plan(plan_forward(B, He, V, He, false), B, He, V) && plane_plan(plan_forward(B, V, He, V, false), B, V, He) &&
           plane_plan(plan_stats(V, He, 2 * B, He, false), V, He, 2 * B);
}

bool planes_eligible(const mdbn_cd_args* a)
{
    if (!a->planes || !a->W_planes) return false;
    if (a->persistent || a->sample_stats || (a->gauss && a->add_noise)) return false;
    const int64_t B = a->B, V = a->V, H = a->H;
    if (!plane_shape_ok(B, V, H, a->ldv, a->ldh)) return false;
    return a->planes_bytes >= 2 * planes_elems(B, a->ldv, a->ldh) && aligned16(a->planes) && aligned16(a->W_planes);
}

// Workgroups of a balanced launch over `units` stage units (0: launch one workgroup per tile job as usual).  Balanced
// launches exist for the 16x16x32 shape only, need >= 4 stages per workgroup to keep the LDS ring busy, and the in-place
// variant parks through a workspace sized for kMaxBalBlocks workgroups.
int bal_blocks(const mdbn_ctx* ctx, int comm_cus, int64_t tiles, int64_t stages)
{
    if (comm_cus <= 0) comm_cus = g_opt_comm_cus;
    if ((comm_cus <= 0 && g_opt_bal_blocks <= 0) || g_opt_planes_mfma != 16) return 0;
    int P = g_opt_bal_blocks > 0 ? g_opt_bal_blocks : std::min(ctx->num_cu - comm_cus, kMaxBalBlocks);
    if (P < 1) return 0;
    // Fewer tiles than workgroups (forward passes): every tile is shared.  When P is not a multiple of the tile count the
    // shares begin at stages scattered all over the reduction index, the workgroups stop sweeping it in step and the
    // operand stages stop being shared in L2 (measured at c2: propup 31 -> 40 us at P = 216, 31 again at P = 192).  An
    // equal number of workgroups per tile keeps them in step; take it unless it costs more than a quarter in stages.
    if (tiles < P && P % tiles != 0 && g_opt_bal_blocks <= 0) {
        const int aligned = (int)(tiles * (P / tiles));
        if (4ll * P <= 5ll * aligned) P = aligned;          // stages per workgroup grow by P / aligned <= 1.25
    }
    return tiles * stages >= 4 * (int64_t)P ? P : 0;
}

// One forward pass on planes: A planes [rows, K] (ROW), W planes as COL (dir 0: propup) or ROW (dir 1: propdown);
// `e` arrives with outputs / bias / rng / colsum set, this fills in the slab side and the cost partials.
// (H: the hidden width the GEMMs run on -- ldh, see plane_shape_ok; ncols: the live columns of the output)
int run_affine_planes(mdbn_ctx* ctx, int comm_cus, const unsigned short* A, int64_t lda, int64_t pa, int ap, int dir,
                      const unsigned short* Wp, int64_t V, int64_t H, int64_t ncols, int64_t rows, EpiArgs e, bool want_cost,
                      const Workspace& ws, hipStream_t s, int* n_cost_out)
{
    const int64_t Kdim = dir == 0 ? V : H, Ndim = dir == 0 ? H : V;
    PlaneGemmArgs g{};
    g.A = A; g.lda = lda; g.pa = pa; g.ap = ap;
    g.B = Wp; g.ldb = H; g.pb = V * H;
    g.M = (int)rows; g.N = (int)Ndim; g.K = (int)Kdim;
    g.tiles_m = (int)(rows / 128); g.tiles_n = (int)(Ndim / 128);
    {
        const Plan p = plan_forward(rows, Ndim, Kdim, e.ld, false);        // eligibility checked that this is a whole-tile x6 plan
        g.splitk = p.splitk; g.kchunk = p.kchunk;
    }
    // data-parallel mode: P workgroups share tiles x stages evenly; a tile's segments land in slabs, the epilogue launch
    // sums the slabs each tile has
    int bal = bal_blocks(ctx, comm_cus, (int64_t)g.tiles_m * g.tiles_n, Kdim / 32), bal_slabs = 0;
    if (bal) {
        bal_slabs = bal_max_segments(g.tiles_m * g.tiles_n, (int)(Kdim / 32), bal);
        if ((int64_t)bal_slabs * rows * e.ld > ws.slab_floats) bal = 0;        // a workspace sized before the option was set
        if ((int64_t)bal_slabs * rows * e.ld * 4 >= (int64_t)1 << 31) bal = 0;  // 32-bit buffer offsets
    }
    // 128 x 64 tiles instead of a 2-way split-K (propdown at c2: 256 tiles, the activation on the parked tile, no slabs and
    // no epilogue launch): "narrow_tiles" (default on), 16x16x32 shape, ROW W operand, a split factor of exactly 2
    if (!bal && g_opt_narrow_tiles && g_opt_fused_epilogue && dir == 1 && g.splitk == 2 && Ndim % 64 == 0 && g_opt_planes_mfma == 16 &&
        !g_opt_bf16_inputs && (ap == 1 || ap == 3) && (int64_t)g.tiles_m * (Ndim / 64) <= 2 * ctx->num_cu &&
        (!e.target || (e.ld_target % 4 == 0 && ((uintptr_t)e.target & 15) == 0))) {      // its epilogue reads targets as float4
        g.bn = 64; g.tiles_n = (int)(Ndim / 64); g.splitk = 1; g.kchunk = (int)Kdim;
    }
    const bool fuse = !bal && g_opt_fused_epilogue && g.splitk == 1;
    const int nb = fuse ? g.tiles_m * g.tiles_n : epilogue_blocks(rows, e.ld);
    e.rows = (int)rows; e.cols = (int)ncols;
    e.cost_partials = nullptr;
    if (want_cost) {
        if (nb > ws.cost_floats) return fail(MDBN_ENOSPC, "cost scratch exhausted");
        e.cost_partials = ws.cost_partials;
        if (n_cost_out) *n_cost_out = nb;
    }
    const int lb = dir == 0 ? LAY_MN : LAY_K;
    if (bal) {
        g.bal = bal; g.fused = 0; g.C = ws.slabs; g.ldc = e.ld; g.slab_stride = rows * e.ld;
        // the shares of one tile have no operand stage in common; tiles in one row / column of tiles do.  With many more
        // workgroups than tiles deal a tile's shares over the XCDs, otherwise keep neighbouring tiles on one XCD
        // (c2 data-parallel step at P = 224: 190.9 us with this rule, 194.6 all dealt, 193.8 all grouped)
        g.xcd_group = bal >= 4 * g.tiles_m * g.tiles_n ? 0 : 1;
        g.splitk = 1; g.kchunk = (int)Kdim; g.c_bytes = (int64_t)bal_slabs * rows * e.ld * 4;
        e.slabs = ws.slabs; e.slab_stride = g.slab_stride; e.nsplit = bal_slabs;
        e.bal_P = bal; e.bal_S = (int)(Kdim / 32); e.bal_tiles_m = g.tiles_m; e.bal_tiles_n = g.tiles_n;
        {   // every tile with the same number of slabs: the epilogue needs no per-tile count (and has unrolled variants)
            const int tiles = g.tiles_m * g.tiles_n;
            bool uniform = true;
            for (int t = 0; t < tiles && uniform; ++t) uniform = bal_tile_slabs(t, tiles, e.bal_S, bal) == bal_slabs;
            if (uniform) e.bal_P = 0;
        }
        HIP_OK(timed_gemm_planes(LAY_K, lb, g, s));
        HIP_OK(launch_act_epilogue(e, s));
    } else if (fuse) {
        g.fused = 1; g.epi = e;
        HIP_OK(timed_gemm_planes(LAY_K, lb, g, s));
    } else {
        REQUIRE((int64_t)g.splitk * rows * e.ld <= ws.slab_floats, "internal: plane GEMM slabs exceed the workspace");
        g.fused = 0; g.C = ws.slabs; g.ldc = e.ld; g.slab_stride = rows * e.ld;
        e.slabs = ws.slabs; e.slab_stride = g.slab_stride; e.nsplit = g.splitk;
        HIP_OK(timed_gemm_planes(LAY_K, lb, g, s));
        HIP_OK(launch_act_epilogue(e, s));
    }
    return MDBN_OK;
}

// The CD-k step on planes (same sequence, draws and outputs as cd_step_impl below; GRBM without noise and
// Bernoulli RBM, CD only).  upd != NULL: single-device step with the update fused into the statistics GEMM.
// mode: 0 = the whole step, 1 = everything before the statistics GEMM (mdbn_cd_forward), 2 = the statistics GEMM only
// (mdbn_cd_statistics).  defer (modes 0 / 2, upd == NULL): the previous step's deferred update (phase 3), applied by the
// statistics GEMM's loader waves when it qualifies, else launched as update_kernel right before that GEMM.
int cd_step_planes(mdbn_ctx* ctx, hipStream_t s, const mdbn_cd_args* a, const mdbn_update_args* upd, const Workspace& ws,
                   int mode = 0, const mdbn_update_args* defer = nullptr)
{
    const int64_t B = a->B, V = a->V, Hlive = a->H, ldv = V, ldh = a->ldh;
    const int64_t H = ldh;          // the width the GEMMs run on (pad columns: exact zeros; plane_shape_ok)
    PlaneBufs pb;
    unsigned short* Xother = nullptr;        // the X2 buffer this step does NOT use (gather-ahead target)
    {
        unsigned short* p = reinterpret_cast<unsigned short*>(a->planes);
        pb.Xp = p; pb.px = 2 * B * ldv; p += 6 * B * ldv;
        pb.Pp = p; pb.pp = 2 * B * ldh; p += 6 * B * ldh;
        pb.hsp = p; p += B * ldh;
        pb.vsp = p;
        if (a->planes_alt) {
            REQUIRE(aligned16(a->planes_alt), "planes_alt not 16-byte aligned");
            unsigned short* alt = reinterpret_cast<unsigned short*>(a->planes_alt);
            if (a->x_buffer) { Xother = pb.Xp; pb.Xp = alt; } else Xother = alt;
        } else {
            REQUIRE(a->x_buffer == 0, "x_buffer = 1 needs planes_alt");
        }
    }
    if (a->ahead_done) *a->ahead_done = 0;
    unsigned short* Wp = reinterpret_cast<unsigned short*>(a->W_planes);    // (split on entry if stale: cd_step_impl)
    // float32 copies nobody on the path reads (the GEMMs take planes, the bias statistics their column partials)
    const bool keep = a->keep_f32 != 0 || a->trace_h != nullptr || a->trace_v != nullptr;

    float* v0 = a->V2;
    float* nv = a->V2 + B * ldv;
    float* ph = a->P2;
    float* nh = a->P2 + B * ldh;
    // x = train_set_x[indexes] (dbn.py:307), as f32 (cost target, bias statistics) and as planes
    // (without `keep` the float32 copy of v0 is not made either: its one reader, the reconstruction-cost / bias-statistics
    // target of the last visible pass, reads the dataset rows through the index instead)
    int n_cost = 0;
    if (mode != 2) {
    // (gathered ahead by the previous call's statistics kernel: the planes are already there)
    if (!(a->v0_ready && !keep))
        HIP_OK(launch_gather_planes(a->data, a->n_data, ldv, ldv, a->indexes, a->index_is_64, B, keep ? v0 : nullptr, ldv, pb.Xp,
                                    pb.px, s));

    auto key = [&](uint32_t draw) { PhiloxKey k = make_key(a->rng, draw); return k; };
    {   // positive phase: ph_mean (+ planes), h0 sample (f32 for the taps, plane for the chain)   (rbm.py:303)
        EpiArgs e{};
        e.ld = ldh; e.bias = a->hbias; e.mean = keep ? ph : nullptr; e.sample = keep ? a->hs : nullptr; e.mean_scale = 1.0f; e.gauss = 0;
        e.colsum = ws.colPpos; e.colsum_kind = 0; e.rng = key(0);
        e.mean_planes = pb.Pp; e.plane_stride = pb.pp; e.sample_plane = pb.hsp;
        CHECK(run_affine_planes(ctx, a->comm_cus, pb.Xp, ldv, pb.px, 3, 0, Wp, V, H, Hlive, B, e, false, ws, s, nullptr));
        if (a->trace_h) HIP_OK(hipMemcpyAsync(a->trace_h, a->hs, sizeof(float) * B * ldh, hipMemcpyDeviceToDevice, s));
    }
    for (int t = 1; t <= a->k; ++t) {                                  // gibbs_hvh x k (rbm.py:318-336)
        const bool last = t == a->k;
        {   // v_t | h_{t-1}: the chain state is our own 0/1 sample: one plane, three products
            EpiArgs e{};
            e.ld = ldv; e.bias = a->vbias; e.mean = keep ? nv : nullptr; e.mean_scale = 1.0f; e.gauss = a->gauss;
            e.sample = (a->gauss || !keep) ? nullptr : a->vs; e.rng = key((uint32_t)(2 * t - 1));
            e.mean_planes = pb.Xp + B * ldv; e.plane_stride = pb.px;      // rows B..2B-1 of the X2 planes
            e.sample_plane = a->gauss ? nullptr : pb.vsp;
            if (last) {
                e.colsum = ws.colV; e.colsum_kind = 1;
                if (keep) { e.target = v0; e.ld_target = ldv; }
                else {
                    e.target = a->data; e.ld_target = ldv; e.target_rows = a->n_data;
                    e.target_idx = a->indexes; e.target_idx64 = a->index_is_64;       // NULL: rows 0..B-1 of the data
                }
            }
            CHECK(run_affine_planes(ctx, a->comm_cus, pb.hsp, ldh, B * ldh, 1, 1, Wp, V, H, V, B, e, last, ws, s, last ? &n_cost : nullptr));
            if (a->trace_v && !a->gauss)
                HIP_OK(hipMemcpyAsync(a->trace_v + (int64_t)(t - 1) * B * ldv, a->vs, sizeof(float) * B * ldv,
                                      hipMemcpyDeviceToDevice, s));
        }
        {   // h_t | v_t: from the mean for GRBM (rbm.py:669), from the 0/1 sample for RBM (rbm.py:246)
            const bool need_sample = !last;
            EpiArgs e{};
            e.ld = ldh; e.bias = a->hbias; e.mean = keep ? nh : nullptr; e.mean_scale = -1.0f; e.gauss = 0;
            e.sample = (need_sample && keep) ? a->hs : nullptr; e.rng = key((uint32_t)(2 * t));
            e.mean_planes = pb.Pp + B * ldh; e.plane_stride = pb.pp;       // rows B..2B-1 of the P2 planes: -nh_mean
            e.sample_plane = need_sample ? pb.hsp : nullptr;
            if (last) { e.colsum = ws.colPneg; e.colsum_kind = 0; }
            if (a->gauss) CHECK(run_affine_planes(ctx, a->comm_cus, pb.Xp + B * ldv, ldv, pb.px, 3, 0, Wp, V, H, Hlive, B, e, false, ws, s, nullptr));
            else CHECK(run_affine_planes(ctx, a->comm_cus, pb.vsp, ldv, B * ldv, 1, 0, Wp, V, H, Hlive, B, e, false, ws, s, nullptr));
            if (a->trace_h && need_sample)
                HIP_OK(hipMemcpyAsync(a->trace_h + (int64_t)t * B * ldh, a->hs, sizeof(float) * B * ldh, hipMemcpyDeviceToDevice, s));
        }
    }

    }       // mode != 2
    if (mode == 1) {
        ctx->pending_n_cost = n_cost; ctx->pending_stats = a->stats;
        return MDBN_OK;
    }
    if (mode == 2) n_cost = ctx->pending_n_cost;

    float* S = a->stats;
    float* s_h = a->stats + V * ldh;
    float* s_v = s_h + ldh;
    float* cost = s_v + ldv;
    // S = [v0; nv]^T [ph; -nh]: one GEMM over the stacked batch dimension, both operands used transposed
    const Plan sp = plan_stats(V, H, 2 * B, ldh, false);
    PlaneGemmArgs g{};
    g.A = pb.Xp; g.lda = ldv; g.pa = pb.px; g.ap = 3;
    g.B = pb.Pp; g.ldb = ldh; g.pb = pb.pp;
    g.M = (int)V; g.N = (int)H; g.K = (int)(2 * B);
    g.tiles_m = (int)(V / 128); g.tiles_n = (int)(H / 128); g.splitk = sp.splitk; g.kchunk = sp.kchunk;
    g.fin_enabled = 0;
    // data-parallel mode: the statistics land in place from a balanced launch (partials of shared tiles go through
    // scratch carved from the slab region, which the forward passes no longer need)
    int bal = sp.splitk == 1 ? bal_blocks(ctx, a->comm_cus, (int64_t)g.tiles_m * g.tiles_n, 2 * B / 32) : 0;
    if (bal && ((int64_t)bal * 2 * 16384 > ws.slab_floats || V * ldh * 4 >= (int64_t)1 << 31)) bal = 0;
    const bool fuse_upd = !bal && upd != nullptr && g_opt_fused_update && sp.splitk == 1;
    // gather-ahead of the next minibatch by the loader waves, after their W chunks (g.upd.early set): one 256-octet pass of
    // one row per stage; every workgroup takes rpw consecutive rows
    auto set_gather_ahead = [&]() {
        if (g.upd.early && g_opt_gather_ahead && a->next_indexes && Xother && !keep) {
            const int nwg = g.tiles_m * g.tiles_n, nt = (int)(2 * B / 32);
            const int rpw = (int)((B + nwg - 1) / nwg), passes = (int)((ldv / 8 + 255) / 256);
            if (rpw * passes <= 4 && 16 / (nt >= 20 ? 1 : 2) + 4 + 1 <= nt - 3) {      // (the kernel runs four unit slots, always)
                g.ga.src = a->data; g.ga.n_rows = a->n_data; g.ga.ld_src = ldv;
                g.ga.idx = a->next_indexes; g.ga.idx64 = a->index_is_64;
                g.ga.B = (int)B; g.ga.rpw = rpw; g.ga.passes = passes;
                g.ga.P = Xother; g.ga.plane_stride = pb.px; g.ga.ld = ldv;
                if (a->ahead_done) *a->ahead_done = 1;
            }
        }
    };
    if (fuse_upd) {
        BiasUpd bu;
        bu.hb = upd->hbias; bu.hbs = upd->hbias_speed; bu.vb = upd->vbias; bu.vbs = upd->vbias_speed;
        bu.H = Hlive; bu.V = V; bu.lr = upd->lr; bu.mu = upd->momentum; bu.inv_rows = 1.0f / upd->n_rows;
        bu.cost_scale = upd->cost_scale; bu.cost_out = upd->cost_out;
        g.fused = 2;
        g.fin_enabled = 1;
        g.fin = make_fin_args(ws.colPpos, ws.colPneg, ws.colV, row_groups(B), ldh, ldv, ws.cost_partials, n_cost, s_h, s_v,
                              cost, &bu);
        g.upd.W = upd->W; g.upd.Ws = upd->W_speed; g.upd.W0 = upd->W0; g.upd.ld = ldh; g.upd.rows = (int)V;
        g.upd.lr = upd->lr; g.upd.l1 = upd->lambda_1; g.upd.l2 = upd->lambda_2; g.upd.wc = upd->weightcost;
        g.upd.mu = upd->momentum; g.upd.inv_bs = 1.0f / upd->batch_size;
        g.upd.Wp = Wp; g.upd.wp_stride = V * ldh;
        // parameter half applied by the loader waves during the main loop (mdbn_planes.hip, EARLYW): needs the split-phase
        // conditions of the update (no lambda_1; weight cost off or on a frozen snapshot) and >= 12 stages to spread over
        g.upd.early = g_opt_early_w && upd->lambda_1 == 0.f && (upd->weightcost == 0.f || upd->W0 != nullptr) &&
                      2 * B / 32 >= 12 && g_opt_planes_mfma == 16 && !g_opt_bf16_inputs;
        set_gather_ahead();
        HIP_OK(timed_gemm_planes(LAY_MN, LAY_MN, g, s));
        return MDBN_OK;
    }
    // bias statistics + cost total: inside the statistics GEMM (its MFMA waves run the units while the first stages are
    // in flight) when that GEMM is one launch, a launch of its own (~6 us of dependent tiny kernel) otherwise
    if (sp.splitk == 1 && g_opt_fused_finalize) {
        g.fin_enabled = 1;
        g.fin = make_fin_args(ws.colPpos, ws.colPneg, ws.colV, row_groups(B), ldh, ldv, ws.cost_partials, n_cost, s_h, s_v,
                              cost, nullptr);
    } else {
        HIP_OK(launch_finalize_stats(ws.colPpos, ws.colPneg, ws.colV, row_groups(B), ldh, ldv, ws.cost_partials, n_cost, s_h,
                                     s_v, cost, nullptr, s));
    }
    g.fused = 0; g.ldc = ldh; g.slab_stride = V * ldh;
    mdbn_update_args u;
    if (upd) { u = *upd; u.phase = 0; }
    if (defer) {
        // the previous step's deferred update: inside this GEMM's loader waves (one workgroup per tile, unsplit, >= 20 stages,
        // the split-phase conditions), else as its own launch right here -- bitwise the same either way
        const bool splitphase = g_opt_early_w && g_opt_planes_mfma == 16 && !g_opt_bf16_inputs && defer->phase == 3 &&
                                defer->lambda_1 == 0.f && (defer->weightcost == 0.f || defer->W0 != nullptr) &&
                                defer->stats != a->stats;
        // balanced launch: a flat share of the arrays per workgroup, 16 items of 512 pieces at most, >= 20 stages each
        const int64_t flat = bal ? (((V * ldh / 4 + bal - 1) / bal + 511) / 512) * 512 : 0;
        const bool inside = splitphase && (bal ? (flat <= 16 * 512 && (int64_t)g.tiles_m * g.tiles_n * (2 * B / 32) / bal >= 20)
                                               : (sp.splitk == 1 && 2 * B / 32 >= 20));
        g.upd.flat_per_wg = flat;
        if (inside) {
            g.upd.W = defer->W; g.upd.Ws = defer->W_speed; g.upd.W0 = defer->W0; g.upd.ld = ldh; g.upd.rows = (int)V;
            g.upd.lr = defer->lr; g.upd.l1 = 0.f; g.upd.l2 = defer->lambda_2; g.upd.wc = defer->weightcost;
            g.upd.mu = defer->momentum; g.upd.inv_bs = 1.0f / defer->batch_size;
            g.upd.Wp = reinterpret_cast<unsigned short*>(defer->W_planes); g.upd.wp_stride = V * ldh;
            g.upd.Sprev = defer->stats; g.upd.early = 2;
            const float* ps_h = defer->stats + V * ldh;
            g.db.on = 1; g.db.hb = defer->hbias; g.db.hbs = defer->hbias_speed; g.db.vb = defer->vbias; g.db.vbs = defer->vbias_speed;
            g.db.s_h = ps_h; g.db.s_v = ps_h + ldh; g.db.cost_sum = ps_h + ldh + ldv; g.db.H = Hlive; g.db.V = V;
            g.db.lr = defer->lr; g.db.mu = defer->momentum; g.db.inv_rows = 1.0f / defer->n_rows;
            g.db.cost_scale = defer->cost_scale; g.db.cost_out = defer->cost_out;
            if (!bal) set_gather_ahead();
        } else {
            HIP_OK(launch_update(*defer, s, nullptr, 1, 0, reinterpret_cast<unsigned short*>(defer->W_planes)));
        }
    }
    if (bal) {
        g.xcd_group = 1;
        g.C = S; g.bal = bal; g.fused = 4; g.kchunk = (int)(2 * B); g.c_bytes = V * ldh * 4;
        g.scratch = ws.slabs;
        HIP_OK(timed_gemm_planes(LAY_MN, LAY_MN, g, s));
    } else if (sp.splitk == 1) {
        g.C = S;
        HIP_OK(timed_gemm_planes(LAY_MN, LAY_MN, g, s));
    } else {
        REQUIRE(sp.slab_floats(V, ldh) <= ws.slab_floats, "internal: statistic slabs exceed workspace");
        g.C = ws.slabs;
        HIP_OK(timed_gemm_planes(LAY_MN, LAY_MN, g, s));
        if (upd && g_opt_fused_update) {     // the update sums the slabs itself (same order as sum_slabs_kernel)
            HIP_OK(launch_update(u, s, ws.slabs, sp.splitk, g.slab_stride, Wp));
            return MDBN_OK;
        }
        HIP_OK(launch_sum_slabs(ws.slabs, sp.splitk, g.slab_stride, V * ldh, S, s));
    }
    if (upd) HIP_OK(launch_update(u, s, nullptr, 1, 0, Wp));
    return MDBN_OK;
}

}  // namespace

extern "C" {

int mdbn_version(void) { return MDBN_VERSION; }

#ifndef MDBN_SRC_HASH
#define MDBN_SRC_HASH "unknown"
#endif
// tagged copy: build.py finds the hash of an existing .so by scanning the file's bytes for this tag, without
// dlopen'ing it (a dlopen'ed path stays mapped by name: a rebuild at the same path would not be re-read)
static const char mdbn_source_hash_tag[] __attribute__((used)) = "MDBN_SOURCE_HASH_TAG=" MDBN_SRC_HASH;
int mdbn_source_hash(char* buf, size_t n)
{
    if (!buf || n == 0) return MDBN_EINVAL;
    snprintf(buf, n, "%s", mdbn_source_hash_tag + sizeof("MDBN_SOURCE_HASH_TAG=") - 1);
    return MDBN_OK;
}

int mdbn_last_error(char* buf, size_t n)
{
    if (!buf || n == 0) return MDBN_EINVAL;
    snprintf(buf, n, "%s", g_err.c_str());
    return MDBN_OK;
}

int mdbn_ctx_create(mdbn_ctx** out, int device)
{
    REQUIRE(out != nullptr, "out is NULL");
    int count = 0;
    HIP_OK(hipGetDeviceCount(&count));
    REQUIRE(device >= 0 && device < count, "device %d out of range (%d visible)", device, count);
    hipDeviceProp_t prop;
    HIP_OK(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(MDBN_EHIP, "device %d is %s; this library is built for gfx950 only", device, prop.gcnArchName);
    mdbn_ctx* c = new mdbn_ctx;
    c->device = device;
    c->num_cu = prop.multiProcessorCount;
    *out = c;
    return MDBN_OK;
}

int mdbn_comm_destroy(mdbn_ctx* ctx);

int mdbn_ctx_destroy(mdbn_ctx* ctx)
{
    CtxScope ctx_scope(ctx);
    if (ctx) {
        if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
        if (ctx->ev_join) (void)hipEventDestroy(ctx->ev_join);
        if (ctx->side) (void)hipStreamDestroy(ctx->side);
        if (ctx->comm) (void)mdbn_comm_destroy(ctx);
        if (ctx->gc_flags) (void)hipFree(ctx->gc_flags);
    }
    delete ctx;
    return MDBN_OK;
}

#ifdef MDBN_STAMP
int mdbn_debug_set_stamps(void* p) { g_stamps = (unsigned long long*)p; return MDBN_OK; }
#endif

int mdbn_bal_segment(int32_t tiles, int32_t stages, int32_t workgroups, int32_t w, int32_t k, int32_t* out)
{
    REQUIRE(out != nullptr, "out is NULL");
    int tile, s0, s1, nseg, nt;
    if (bal_segment_host(tiles, stages, workgroups, w, k, &tile, &s0, &s1, &nseg, &nt) != 0)
        return fail(MDBN_EINVAL, "bad arguments");
    out[0] = tile; out[1] = s0; out[2] = s1; out[3] = nseg; out[4] = nt;
    out[5] = tile >= 0 ? bal_tile_slabs(tile, tiles, stages, workgroups) : 0;
    return MDBN_OK;
}

int mdbn_set_option(mdbn_ctx* ctx, const char* name, int64_t value)
{
    CtxScope ctx_scope(ctx);
    REQUIRE(ctx != nullptr && name != nullptr, "NULL argument");
    if (strcmp(name, "gemm_bk") == 0) {
        REQUIRE(value == 0 || value == 32 || value == 64, "gemm_bk must be 0 (auto), 32 or 64");
        ctx->opt.gemm_bk = (int)value;
        return MDBN_OK;
    }
    if (strcmp(name, "epilogue_threads") == 0) {
        if (value != 0 && value != 64 && value != 128 && value != 256) return fail(MDBN_EINVAL, "epilogue_threads must be 0, 64, 128 or 256");
        ctx->opt.epilogue_threads = (int)value;
        return MDBN_OK;
    }
    if (strcmp(name, "epilogue_cw") == 0) {
        REQUIRE(value == 0 || value == 1 || value == 2 || value == 4, "epilogue_cw must be 0 (auto), 1, 2 or 4");
        ctx->opt.epilogue_cw = (int)value;
        return MDBN_OK;
    }
    if (strcmp(name, "skinny_gemm") == 0) {
        ctx->opt.skinny_gemm = value != 0;
        return MDBN_OK;
    }
    if (strcmp(name, "thin_fused") == 0) {
        ctx->opt.thin_fused = value != 0;
        return MDBN_OK;
    }
    if (strcmp(name, "gchain") == 0) {
        ctx->opt.gchain = value != 0;
        return MDBN_OK;
    }
    if (strcmp(name, "skinny_fused_max_k") == 0) {
        ctx->opt.skinny_fused_max_k = value;
        return MDBN_OK;
    }
    if (strcmp(name, "x6_min_jobs") == 0) {
        ctx->opt.x6_min_jobs = (int)value;
        return MDBN_OK;
    }
    if (strcmp(name, "x6_producer_waves") == 0) {
        if (value != 2 && value != 4) return fail(MDBN_EINVAL, "x6_producer_waves must be 2 or 4");
        ctx->opt.x6_pw = (int)value;
        return MDBN_OK;
    }
    if (strcmp(name, "gemm_min_splitk") == 0) {
        if (value < 32) return fail(MDBN_EINVAL, "gemm_min_splitk must be >= 32");
        ctx->opt.min_splitk = (int)value;
        return MDBN_OK;
    }
    if (strcmp(name, "gemm_cw") == 0) {
        if (value < 0 || value > 2) return fail(MDBN_EINVAL, "gemm_cw must be 0 (auto), 1 or 2");
        ctx->opt.gemm_cw = (int)value;
        return MDBN_OK;
    }
    if (strcmp(name, "stream_x6") == 0) {
        ctx->opt.stream_x6 = (int)value;
        return MDBN_OK;
    }
    if (strcmp(name, "stream_max_macs") == 0) {
        ctx->opt.stream_max_macs = value;
        return MDBN_OK;
    }
    if (strcmp(name, "stream_mi") == 0) {
        REQUIRE(value == 0 || value == 1 || value == 2, "stream_mi must be 0, 1 or 2");
        ctx->opt.stream_mi = (int)value;
        return MDBN_OK;
    }
    if (strcmp(name, "stream_ni") == 0) {
        REQUIRE(value == 0 || value == 1 || value == 2, "stream_ni must be 0, 1 or 2");
        ctx->opt.stream_ni = (int)value;
        return MDBN_OK;
    }
    if (strcmp(name, "skinny_max_macs") == 0) {
        ctx->opt.skinny_max_macs = value;
        return MDBN_OK;
    }
    if (strcmp(name, "gemm_bf16x6") == 0) {
        ctx->opt.gemm_bf16x6 = (int)value & 3;
        return MDBN_OK;
    }
    if (strcmp(name, "planes_min_work") == 0) {
        if (value < 0) return fail(MDBN_EINVAL, "planes_min_work must be >= 0");
        ctx->opt.planes_min_work = value;
        return MDBN_OK;
    }
    if (strcmp(name, "bf16_inputs") == 0) {
        ctx->opt.bf16_inputs = value != 0;
        return MDBN_OK;
    }
    if (strcmp(name, "planes_mfma") == 0) {
        if (value != 16 && value != 32) return fail(MDBN_EINVAL, "planes_mfma must be 16 or 32");
        ctx->opt.planes_mfma = (int)value;
        return MDBN_OK;
    }
    if (strcmp(name, "narrow_tiles") == 0) {
        ctx->opt.narrow_tiles = value != 0;
        return MDBN_OK;
    }
    if (strcmp(name, "feed_copy_streams") == 0) {
        REQUIRE(value == 1 || value == 2, "feed_copy_streams must be 1 or 2");
        ctx->opt.feed_copy_streams = value;
        return MDBN_OK;
    }
    if (strcmp(name, "gather_ahead") == 0) {
        ctx->opt.gather_ahead = value != 0;
        return MDBN_OK;
    }
    if (strcmp(name, "early_w") == 0) {
        ctx->opt.early_w = value != 0;
        return MDBN_OK;
    }
    if (strcmp(name, "bal_blocks") == 0) {
        if (value < 0 || value > kMaxBalBlocks) return fail(MDBN_EINVAL, "bal_blocks must be in [0, %d]", kMaxBalBlocks);
        ctx->opt.bal_blocks = (int)value;
        return MDBN_OK;
    }
    if (strcmp(name, "comm_cus") == 0) {
        if (value < 0 || value > 192) return fail(MDBN_EINVAL, "comm_cus must be in [0, 192]");
        ctx->opt.comm_cus = (int)value;
        return MDBN_OK;
    }
    if (strcmp(name, "small_fin_lanes") == 0) {
        REQUIRE(value == 0 || value == 1 || value == 2 || value == 4 || value == 8 || value == 16, "small_fin_lanes must be 0, 1, 2, 4, 8 or 16");
        ctx->opt.small_fin_lanes = (int)value;
        return MDBN_OK;
    }
    if (strcmp(name, "small_fused") == 0) {
        ctx->opt.small_fused = value != 0;
        return MDBN_OK;
    }
    if (strcmp(name, "gemm_planes") == 0) {
        ctx->opt.gemm_planes = value != 0;
        return MDBN_OK;
    }
    if (strcmp(name, "fused_finalize") == 0) {
        ctx->opt.fused_finalize = value != 0;
        return MDBN_OK;
    }
    if (strcmp(name, "fused_update") == 0) {
        ctx->opt.fused_update = value != 0;
        return MDBN_OK;
    }
    if (strcmp(name, "fused_epilogue") == 0) {
        ctx->opt.fused_epilogue = value != 0;
        return MDBN_OK;
    }
    if (strcmp(name, "update_overlap") == 0) {
        ctx->opt.update_overlap = value != 0;
        return MDBN_OK;
    }
    return fail(MDBN_EINVAL, "unknown option %s", name);
}

int mdbn_kernel_timing(mdbn_ctx* ctx, int enable)
{
    CtxScope ctx_scope(ctx);
    REQUIRE(ctx != nullptr, "ctx is NULL");
    g_timing.enabled = enable != 0;
    g_timing.used = 0;
    return MDBN_OK;
}

int mdbn_kernel_timing_read(mdbn_ctx* ctx, int64_t* n_launches, double* total_ms)
{
    CtxScope ctx_scope(ctx);
    REQUIRE(ctx && n_launches && total_ms, "NULL argument");
    double tot = 0.0;
    for (size_t i = 0; i < g_timing.used; ++i) {
        HIP_OK(hipEventSynchronize(g_timing.pool[i].second));
        float ms = 0.f;
        HIP_OK(hipEventElapsedTime(&ms, g_timing.pool[i].first, g_timing.pool[i].second));
        tot += ms;
    }
    *n_launches = (int64_t)g_timing.used;
    *total_ms = tot;
    return MDBN_OK;
}

int mdbn_kernel_timing_detail(mdbn_ctx* ctx, int64_t cap, double* ms, double* alg_flop, double* pipe_flop,
                              int32_t* kind, int64_t* n_launches)
{
    CtxScope ctx_scope(ctx);
    REQUIRE(ctx && n_launches && (cap == 0 || (ms && alg_flop && pipe_flop && kind)), "NULL argument");
    const int64_t n = std::min<int64_t>(cap, (int64_t)g_timing.used);
    for (int64_t i = 0; i < n; ++i) {
        HIP_OK(hipEventSynchronize(g_timing.pool[i].second));
        float t = 0.f;
        HIP_OK(hipEventElapsedTime(&t, g_timing.pool[i].first, g_timing.pool[i].second));
        ms[i] = t;
        alg_flop[i] = g_timing.meta[i].alg_flop;
        pipe_flop[i] = g_timing.meta[i].pipe_flop;
        kind[i] = g_timing.meta[i].kind;
    }
    *n_launches = (int64_t)g_timing.used;
    return MDBN_OK;
}

int mdbn_workspace_bytes(int64_t B, int64_t V, int64_t H, int64_t* bytes)
{
    REQUIRE(bytes != nullptr && B > 0 && V > 0 && H > 0, "bad arguments");
    *bytes = ws_sizes(B, V, H).total_bytes();
    return MDBN_OK;
}

int mdbn_workspace_bytes_ctx(mdbn_ctx* ctx, int64_t B, int64_t V, int64_t H, int64_t* bytes)
{
    CtxScope ctx_scope(ctx);
    REQUIRE(ctx != nullptr, "ctx is NULL");
    return mdbn_workspace_bytes(B, V, H, bytes);
}

int mdbn_padded_ld(int64_t cols, int64_t* ld)
{
    REQUIRE(ld != nullptr && cols > 0, "bad arguments");
    *ld = padded_ld(cols);
    return MDBN_OK;
}

int mdbn_planes_bytes(int64_t B, int64_t ldv, int64_t ldh, int64_t* bytes)
{
    REQUIRE(bytes != nullptr && B > 0 && ldv > 0 && ldh > 0, "bad arguments");
    *bytes = 2 * planes_elems(B, ldv, ldh);
    return MDBN_OK;
}

int mdbn_planes_eligible(int64_t B, int64_t V, int64_t H, int64_t ldv, int64_t ldh, int32_t* eligible)
{
    REQUIRE(eligible != nullptr, "eligible is NULL");
    *eligible = plane_shape_ok(B, V, H, ldv, ldh) ? 1 : 0;
    return MDBN_OK;
}

int mdbn_planes_eligible_ctx(mdbn_ctx* ctx, int64_t B, int64_t V, int64_t H, int64_t ldv, int64_t ldh, int32_t* eligible)
{
    CtxScope ctx_scope(ctx);
    REQUIRE(ctx != nullptr, "ctx is NULL");
    return mdbn_planes_eligible(B, V, H, ldv, ldh, eligible);
}

int mdbn_planes_alt_bytes(int64_t B, int64_t ldv, int64_t* bytes)
{
    REQUIRE(bytes != nullptr && B > 0 && ldv > 0, "bad arguments");
    *bytes = 2 * 6 * B * ldv;
    return MDBN_OK;
}

int mdbn_ahead_bytes_ctx(mdbn_ctx* ctx, int64_t B, int64_t V, int64_t H, int64_t ldv, int64_t ldh, int64_t* bytes)
{
    CtxScope ctx_scope(ctx);
    REQUIRE(ctx != nullptr && bytes != nullptr && B > 0 && V > 0 && H > 0 && ldv >= V && ldh >= H, "bad arguments");
    int64_t n = 2 * 6 * B * ldv;                        // plane path: the second X2-plane buffer
    ThinGeom tg;
    if (thin_geom(B, V, H, ldv, ldh, std::min(ctx->num_cu, kTargetJobs), tg) && tg.lds_ahead > 0)
        n = std::max<int64_t>(n, (int64_t)tg.G * tg.Bq * ldh * 4);     // thin path: the partials of the next positive phase
    *bytes = n;
    return MDBN_OK;
}

int mdbn_split_planes(mdbn_ctx* ctx, void* stream, const float* x, int64_t rows, int64_t ld, void* planes)
{
    CtxScope ctx_scope(ctx);
    REQUIRE(ctx && x && planes && rows > 0 && ld > 0 && ld % 4 == 0, "bad arguments");
    REQUIRE(aligned16(x) && aligned16(planes), "x and planes must be 16-byte aligned");
    HIP_OK(launch_split_planes(x, rows, ld, reinterpret_cast<unsigned short*>(planes), rows * ld, (hipStream_t)stream));
    return MDBN_OK;
}

int mdbn_stats_floats(int64_t V, int64_t ldv, int64_t ldh, int64_t* n)
{
    REQUIRE(n != nullptr && V > 0 && ldv >= V && ldh > 0 && ldv % 4 == 0 && ldh % 4 == 0, "bad arguments");
    *n = V * ldh + ldh + ldv + 4;
    return MDBN_OK;
}

int mdbn_gather_rows(mdbn_ctx* ctx, void* stream, const float* src, int64_t n_rows, int64_t cols,
                     int64_t ld_src, const void* indexes, int index_is_64, int64_t n_idx, float* dst,
                     int64_t ld_dst)
{
    CtxScope ctx_scope(ctx);
    REQUIRE(ctx != nullptr, "ctx is NULL");
    CHECK(check_mat(src, ld_src, cols, "src"));
    CHECK(check_mat(dst, ld_dst, cols, "dst"));
    REQUIRE(n_rows > 0 && n_idx >= 0, "bad row counts");
    REQUIRE(indexes != nullptr || n_idx <= n_rows, "identity gather longer than the source");
    HIP_OK(launch_gather(src, n_rows, ru4(cols), ld_src, indexes, index_is_64, n_idx, dst, ld_dst,
                         (hipStream_t)stream));
    return MDBN_OK;
}

int mdbn_gather_rows_host(mdbn_ctx* ctx, void* stream, const float* src, int64_t n_rows, int64_t cols, int64_t ld_src,
                          const void* indexes, int index_is_64, int64_t n_idx, float* dst, int64_t ld_dst, int workgroups,
                          int threads)
{
    CtxScope ctx_scope(ctx);
    REQUIRE(ctx != nullptr, "ctx is NULL");
    CHECK(check_mat(src, ld_src, cols, "src"));
    CHECK(check_mat(dst, ld_dst, cols, "dst"));
    REQUIRE(n_rows > 0 && n_idx >= 0, "bad row counts");
    REQUIRE(indexes != nullptr || n_idx <= n_rows, "identity gather longer than the source");
    REQUIRE(workgroups >= 0 && workgroups <= 1024, "workgroups must be in [0, 1024] (0 = default 32)");
    REQUIRE(threads == 0 || threads == 64 || threads == 128 || threads == 256, "threads must be 0 (= 256), 64, 128 or 256");
    HIP_OK(launch_gather_slim(src, n_rows, ru4(cols), ld_src, indexes, index_is_64, n_idx, dst, ld_dst,
                              workgroups ? workgroups : 32, threads ? threads : 256, (hipStream_t)stream));
    return MDBN_OK;
}

int mdbn_propup_sample(mdbn_ctx* ctx, void* stream, const float* v, int64_t B, int64_t ldv, const float* W,
                       int64_t V, int64_t H, int64_t ldh, const float* hbias, float* pre, float* mean,
                       float mean_scale, float* sample, const mdbn_rng* rng, void* workspace,
                       int64_t workspace_bytes)
{
    CtxScope ctx_scope(ctx);
    REQUIRE(ctx != nullptr, "ctx is NULL");
    REQUIRE(B > 0 && V > 0 && H > 0, "bad shape");
    CHECK(check_mat(v, ldv, V, "v"));
    CHECK(check_mat(W, ldh, H, "W"));
    REQUIRE(hbias != nullptr, "hbias is NULL");
    REQUIRE(sample == nullptr || rng != nullptr, "sampling needs rng");
    for (const float* p : {(const float*)pre, (const float*)mean, (const float*)sample})
        REQUIRE(p == nullptr || aligned16(p), "output not 16-byte aligned");
    Workspace ws;
    CHECK(carve(workspace, workspace_bytes, B, V, H, ws, false));
    Affine a{v, B, ldv, W, V, H, ldh, 0, hbias, pre, mean, sample, ldh, mean_scale, 0,
             nullptr, 0, false, rng, rng ? rng->draw : 0u};
    return run_affine(a, ws, (hipStream_t)stream, nullptr);
}

int mdbn_propdown_sample(mdbn_ctx* ctx, void* stream, const float* h, int64_t B, int64_t ldh, const float* W,
                         int64_t V, int64_t H, int64_t ldv, const float* vbias, int gauss, int add_noise,
                         float* pre, float* mean, float* sample, const mdbn_rng* rng, const float* v0,
                         float* cost_sum, void* workspace, int64_t workspace_bytes)
{
    CtxScope ctx_scope(ctx);
    REQUIRE(ctx != nullptr, "ctx is NULL");
    REQUIRE(B > 0 && V > 0 && H > 0, "bad shape");
    CHECK(check_mat(h, ldh, H, "h"));
    CHECK(check_mat(W, ldh, H, "W"));
    REQUIRE(vbias != nullptr, "vbias is NULL");
    REQUIRE(ldv % 4 == 0 && ldv >= V, "bad ldv");
    const bool draws = sample != nullptr && (!gauss || add_noise);
    REQUIRE(!draws || rng != nullptr, "sampling needs rng");
    REQUIRE((v0 == nullptr) == (cost_sum == nullptr), "v0 and cost_sum go together");
    Workspace ws;
    CHECK(carve(workspace, workspace_bytes, B, V, H, ws, false));
    hipStream_t s = (hipStream_t)stream;
    // GRBM without noise: sample == mean (rbm.py:652-653): write the mean twice, no draw
    Affine a{h, B, ldh, W, V, H, ldh, 1, vbias, pre, mean, draws ? sample : nullptr, ldv, 1.0f, gauss,
             v0, ldv, v0 != nullptr, rng, rng ? rng->draw : 0u};
    int n_cost = 0;
    CHECK(run_affine(a, ws, s, &n_cost));
    if (sample && !draws) {
        REQUIRE(mean != nullptr || pre != nullptr, "noise-free GRBM sample needs mean or pre");
        if (sample != mean)
            HIP_OK(hipMemcpyAsync(sample, mean ? mean : pre, sizeof(float) * B * ldv, hipMemcpyDeviceToDevice, s));
    }
    if (cost_sum)
        HIP_OK(launch_finalize_stats(nullptr, nullptr, nullptr, 0, 0, 0, ws.cost_partials, n_cost, nullptr,
                                     nullptr, cost_sum, nullptr, s));
    return MDBN_OK;
}

// ---------------------------------------------------------------------------------- RCCL (data parallelism)
// librccl is opened lazily: the library loads and every single-device entry point works without it.
namespace {
struct Rccl {
    void* h = nullptr;
    int (*GetUniqueId)(void*) = nullptr;
    void* CommInitRank = nullptr;          // ncclCommInitRank(ncclComm_t*, int, ncclUniqueId BY VALUE, int)
    int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};
Rccl g_rccl;
struct NcclId { char internal[128]; };

int rccl_load()
{
    if (g_rccl.h) return MDBN_OK;
    void* h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return fail(MDBN_EHIP, "cannot open librccl.so: %s", dlerror());
    g_rccl.GetUniqueId = reinterpret_cast<int (*)(void*)>(dlsym(h, "ncclGetUniqueId"));
    g_rccl.CommInitRank = dlsym(h, "ncclCommInitRank");
    g_rccl.AllReduce = reinterpret_cast<int (*)(const void*, void*, size_t, int, int, void*, hipStream_t)>(dlsym(h, "ncclAllReduce"));
    g_rccl.CommDestroy = reinterpret_cast<int (*)(void*)>(dlsym(h, "ncclCommDestroy"));
    g_rccl.GetErrorString = reinterpret_cast<const char* (*)(int)>(dlsym(h, "ncclGetErrorString"));
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.AllReduce || !g_rccl.CommDestroy)
        return fail(MDBN_EHIP, "librccl.so lacks an expected symbol");
    g_rccl.h = h;
    return MDBN_OK;
}
#define RCCL_OK(expr)                                                                          \
    do {                                                                                       \
        int _r = (expr);                                                                       \
        if (_r != 0)                                                                           \
            return fail(MDBN_EHIP, "%s failed: %s", #expr, g_rccl.GetErrorString ? g_rccl.GetErrorString(_r) : "?"); \
    } while (0)
}  // namespace

int mdbn_comm_unique_id(char* id128)
{
    REQUIRE(id128 != nullptr, "id buffer is NULL");
    CHECK(rccl_load());
    RCCL_OK(g_rccl.GetUniqueId(id128));
    return MDBN_OK;
}

int mdbn_comm_init_rank(mdbn_ctx* ctx, const char* id128, int nranks, int rank)
{
    CtxScope ctx_scope(ctx);
    REQUIRE(ctx != nullptr && id128 != nullptr && nranks >= 1 && rank >= 0 && rank < nranks, "bad arguments");
    REQUIRE(ctx->comm == nullptr, "this context already has a communicator");
    CHECK(rccl_load());
    HIP_OK(hipSetDevice(ctx->device));
    typedef int (*init_fn)(void**, int, NcclId, int);          // ncclUniqueId is passed BY VALUE
    NcclId id;
    memcpy(id.internal, id128, 128);
    RCCL_OK(reinterpret_cast<init_fn>(g_rccl.CommInitRank)(&ctx->comm, nranks, id, rank));
    ctx->comm_ranks = nranks;
    return MDBN_OK;
}

int mdbn_allreduce_stats(mdbn_ctx* ctx, void* stream, float* stats, int64_t n)
{
    CtxScope ctx_scope(ctx);
    REQUIRE(ctx != nullptr && ctx->comm != nullptr, "no communicator: call mdbn_comm_init_rank first");
    REQUIRE(stats != nullptr && n > 0, "bad arguments");
    // in place, float32 (ncclFloat32 = 7), sum (ncclSum = 0): the packed [S | s_h | s_v | cost] buffer of one CD step
    RCCL_OK(g_rccl.AllReduce(stats, stats, (size_t)n, 7, 0, ctx->comm, (hipStream_t)stream));
    return MDBN_OK;
}

int mdbn_comm_destroy(mdbn_ctx* ctx)
{
    CtxScope ctx_scope(ctx);
    REQUIRE(ctx != nullptr, "ctx is NULL");
    if (ctx->comm) {
        RCCL_OK(g_rccl.CommDestroy(ctx->comm));
        ctx->comm = nullptr;
        ctx->comm_ranks = 0;
    }
    return MDBN_OK;
}

int mdbn_gibbs_chain(mdbn_ctx* ctx, void* stream, float* v, int64_t B, int64_t ldv, const float* W, int64_t V, int64_t H,
                     int64_t ldh, const float* hbias, const float* vbias, int gauss, int add_noise, int64_t n_steps,
                     float* pre_h, float* h_mean, float* h_sample, float* pre_v, float* v_mean, const mdbn_rng* rng,
                     void* workspace, int64_t workspace_bytes)
{
    CtxScope ctx_scope(ctx);
    REQUIRE(ctx != nullptr && rng != nullptr, "NULL argument");
    REQUIRE(B > 0 && V > 0 && H > 0 && n_steps >= 1, "bad shape / step count");
    CHECK(check_mat(v, ldv, V, "v"));
    CHECK(check_mat(W, ldh, H, "W"));
    CHECK(check_mat(h_mean, ldh, H, "h_mean"));
    CHECK(check_mat(h_sample, ldh, H, "h_sample"));
    CHECK(check_mat(v_mean, ldv, V, "v_mean"));
    REQUIRE(hbias && vbias, "bias pointers are NULL");
    REQUIRE((pre_h == nullptr || aligned16(pre_h)) && (pre_v == nullptr || aligned16(pre_v)), "pre outputs not aligned");
    Workspace ws;
    CHECK(carve(workspace, workspace_bytes, B, V, H, ws, false));
    hipStream_t s = (hipStream_t)stream;
    const bool noisy = gauss && add_noise;
    for (int64_t t = 0; t < n_steps; ++t) {
        const bool last = t + 1 == n_steps;
        mdbn_rng rh = *rng, rv = *rng;
        rh.step = rng->step + (uint32_t)(2 * t);     rh.draw = 0;      // what 2 * n_steps eager sample_* calls would use
        rv.step = rng->step + (uint32_t)(2 * t + 1); rv.draw = 0;
        // h | v: RBM feeds the SAMPLE on, GRBM the MEAN (rbm.py:253-254, :680)
        const bool need_mean = gauss || last;
        Affine up{v, B, ldv, W, V, H, ldh, 0, hbias, last ? pre_h : nullptr, need_mean ? h_mean : nullptr,
                  (!gauss || last) ? h_sample : nullptr, ldh, 1.0f, 0, nullptr, 0, false, &rh, 0u};
        up.x_binary = !gauss && t > 0;              // from the second step on v holds our own 0/1 samples
        CHECK(run_affine(up, ws, s, nullptr));
        // v | h: the chain state v becomes the visible SAMPLE (RBM: Bernoulli; GRBM: mean, + N(0,1) if noisy)
        Affine down{gauss ? h_mean : h_sample, B, ldh, W, V, H, ldh, 1, vbias, (last && !gauss) ? pre_v : nullptr,
                    (gauss && !noisy) ? v : ((last || noisy) ? v_mean : nullptr), (!gauss || noisy) ? v : nullptr,
                    ldv, 1.0f, gauss, nullptr, 0, false, &rv, 0u};
        down.x_binary = !gauss;
        CHECK(run_affine(down, ws, s, nullptr));
    }
    // noise-free GRBM: sample == mean (rbm.py:652-653); the chain state is the mean itself
    if (gauss && !noisy) HIP_OK(hipMemcpyAsync(v_mean, v, sizeof(float) * B * ldv, hipMemcpyDeviceToDevice, s));
    if (gauss && pre_v) HIP_OK(hipMemcpyAsync(pre_v, v_mean, sizeof(float) * B * ldv, hipMemcpyDeviceToDevice, s));
    return MDBN_OK;
}

int mdbn_cd_stats(mdbn_ctx* ctx, void* stream, const float* V2, const float* P2, int64_t B, int64_t V,
                  int64_t H, int64_t ldv, int64_t ldh, float* stats, void* workspace, int64_t workspace_bytes)
{
    CtxScope ctx_scope(ctx);
    REQUIRE(ctx != nullptr, "ctx is NULL");
    REQUIRE(B > 0 && V > 0 && H > 0, "bad shape");
    CHECK(check_mat(V2, ldv, V, "V2"));
    CHECK(check_mat(P2, ldh, H, "P2"));
    REQUIRE(stats != nullptr && aligned16(stats), "stats must be 16-byte aligned");
    Workspace ws;
    CHECK(carve(workspace, workspace_bytes, B, V, H, ws, true));
    hipStream_t s = (hipStream_t)stream;
    float* S = stats;
    float* s_h = stats + V * ldh;
    float* s_v = s_h + ldh;
    float* cost = s_v + ldv;

    // bias statistics: P2's second half already holds -nh_mean; s_v = sum(v0 - nv_mean)
    const int ng = row_groups(B);
    HIP_OK(launch_colsum_groups(P2, nullptr, (int)B, ldh, ws.colPpos, s));
    HIP_OK(launch_colsum_groups(P2 + B * ldh, nullptr, (int)B, ldh, ws.colPneg, s));
    HIP_OK(launch_colsum_groups(V2, V2 + B * ldv, (int)B, ldv, ws.colV, s));
    HIP_OK(launch_finalize_stats(ws.colPpos, ws.colPneg, ws.colV, ng, ldh, ldv, nullptr, 0, s_h, s_v, cost, nullptr, s));

    // S = [v0 ; nv]^T [ph ; -nh]  : one GEMM over the stacked batch dimension (K = 2B)
    const Plan p = plan_stats(V, H, 2 * B, ldh);
    GemmArgs g{};
    g.A = V2; g.lda = ldv; g.B = P2; g.ldb = ldh;
    g.ldc = ldh; g.slab_stride = V * ldh;
    g.M = (int)V; g.N = (int)H; g.K = (int)(2 * B); g.Nst = (int)ldh;
    p.fill(g);
    g.fused = 0;
    if (p.splitk == 1) {
        g.C = S;
        HIP_OK(timed_gemm(LAY_MN, LAY_MN, g, s));
    } else {
        REQUIRE(p.slab_floats(V, ldh) <= ws.slab_floats, "internal: statistic slabs exceed workspace");
        g.C = ws.slabs;
        HIP_OK(timed_gemm(LAY_MN, LAY_MN, g, s));
        HIP_OK(launch_sum_slabs(ws.slabs, p.splitk, g.slab_stride, V * ldh, S, s));
    }
    return MDBN_OK;
}

static int check_update_args(const mdbn_update_args* a);

int mdbn_apply_update(mdbn_ctx* ctx, void* stream, const mdbn_update_args* a)
{
    CtxScope ctx_scope(ctx);
    REQUIRE(ctx != nullptr && a != nullptr, "NULL argument");
    CHECK(check_update_args(a));
    // phases that write W also rewrite its bf16 planes when the caller keeps some
    HIP_OK(launch_update(*a, (hipStream_t)stream, nullptr, 1, 0,
                         a->phase != 1 ? reinterpret_cast<unsigned short*>(a->W_planes) : nullptr));
    return MDBN_OK;
}

static int check_update_args(const mdbn_update_args* a)
{
    REQUIRE(a->struct_size == sizeof(mdbn_update_args),
            "mdbn_update_args.struct_size is %llu, this library's struct has %llu bytes (built against another mdbn_hip.h?)",
            (unsigned long long)a->struct_size, (unsigned long long)sizeof(mdbn_update_args));
    REQUIRE(a->V > 0 && a->H > 0 && a->ldh >= a->H && a->ldv >= a->V && a->ldh % 4 == 0 && a->ldv % 4 == 0,
            "bad shape / leading dims");
    CHECK(check_mat(a->W, a->ldh, a->H, "W"));
    CHECK(check_mat(a->W_speed, a->ldh, a->H, "W_speed"));
    REQUIRE(a->W0 == nullptr || aligned16(a->W0), "W0 not aligned");
    REQUIRE(a->stats != nullptr && aligned16(a->stats), "stats not aligned");
    REQUIRE(a->hbias && a->hbias_speed && a->vbias && a->vbias_speed, "bias pointers are NULL");
    REQUIRE(a->batch_size > 0.f && a->n_rows > 0.f, "bad divisors");
    REQUIRE(a->phase >= 0 && a->phase <= 3, "phase must be 0, 1, 2 or 3");
    REQUIRE(a->W_planes == nullptr || (aligned16(a->W_planes) && a->ldh % 4 == 0), "W_planes not aligned");
    REQUIRE(a->phase == 0 || (a->lambda_1 == 0.f && (a->weightcost == 0.f || a->W0 != nullptr)),
            "split update phases need lambda_1 == 0 and weightcost == 0 or a frozen W0");
    return MDBN_OK;
}

// ---------------------------------------------------------------------------------- one-launch step (LDS-resident layers)
static bool small_eligible(const mdbn_cd_args* a, const Workspace& ws)
{
    if (!g_opt_small_fused || g_opt_bf16_inputs) return false;
    if (a->persistent || a->sample_stats || (a->gauss && a->add_noise)) return false;
    if (!a->gauss && a->vs == nullptr) return false;
    if (a->B > 65535 * 16 || !small_shape_ok(a->B, a->V, a->H, a->gauss) || !small_ld_ok(a->V, a->H, a->ldv, a->ldh)) return false;
    const int nb = small_blocks(a->B);
    return (int64_t)nb * ((a->V + 63) & ~int64_t(63)) * a->ldh <= ws.slab_floats && (int64_t)nb * SM_NW <= ws.cost_floats && nb <= row_groups(a->B);
}

// mode 0: the whole step; 1: the chain + partials only (mdbn_cd_forward); 2: the finish launch (mdbn_cd_statistics).
// upd: single-device step, the finish launch applies the update; else it stores [S | s_h | s_v | cost] into a->stats.
static int cd_step_small(mdbn_ctx* ctx, hipStream_t s, const mdbn_cd_args* a, const mdbn_update_args* upd, const Workspace& ws,
                         int mode, const mdbn_update_args* defer)
{
    const int64_t B = a->B, V = a->V, H = a->H, ldv = a->ldv, ldh = a->ldh;
    const int nb = small_blocks(B);
    if (mode != 2) {
        SmallCdArgs k{};
        k.data = a->data; k.n_data = a->n_data; k.ld_data = ldv;
        k.idx = a->indexes; k.idx64 = a->index_is_64;
        k.B = (int)B; k.V = (int)V; k.H = (int)H; k.k = a->k; k.gauss = a->gauss;
        k.keep = a->keep_f32 != 0 || a->trace_h || a->trace_v;     // inspection copies of v0 / nv / ph / -nh / samples (as the plane path)
        k.ldv = ldv; k.ldh = ldh;
        k.W = a->W; k.hbias = a->hbias; k.vbias = a->vbias;
        k.rng = make_key(a->rng, 0u);
        k.part_S = ws.slabs; k.posP = ws.colPpos; k.negP = ws.colPneg; k.partV = ws.colV; k.cost_partials = ws.cost_partials;
        k.V2 = a->V2; k.P2 = a->P2; k.hs = a->hs; k.vs = a->vs;
        k.trace_h = a->trace_h; k.trace_v = a->gauss ? nullptr : a->trace_v;
        k.stamps = g_stamps;
        HIP_OK(launch_small_cd(k, s));
    }
    if (mode == 1) {
        ctx->pending_n_cost = nb * SM_NW; ctx->pending_stats = a->stats;
        return MDBN_OK;
    }
    if (mode != 1) { ctx->pending_n_cost = -1; ctx->pending_stats = nullptr; }      // (any step that is not a forward half ends a pending hand-over: the workspace partials are gone)
    // the previous step's deferred update (data-parallel order) is its own launch here, ahead of the finish launch
    if (defer) HIP_OK(launch_update(*defer, s, nullptr, 1, 0, reinterpret_cast<unsigned short*>(defer->W_planes)));
    float* s_h = a->stats + V * ldh;
    float* s_v = s_h + ldh;
    float* cost = s_v + ldv;
    SmallFinArgs f{};
    {
        const SmallLayout L = small_layout((int)V, (int)H, a->gauss != 0);
        f.part = ws.slabs; f.nparts = nb; f.n4p = small_part_quads(L, (int)ldh);
        f.V = (int)V; f.q4 = (int)(ldh >> 2); f.tiles_dn = L.tiles_dn;
    }
    f.S_out = a->stats;
    BiasUpd bu;
    // mdbn_set_option("fused_update", 0): the statistics are materialised and the update is its own launch (update_kernel)
    const bool fuse_upd = upd && g_opt_fused_update;
    if (fuse_upd) {
        f.do_upd = 1;
        f.upd.W = upd->W; f.upd.Ws = upd->W_speed; f.upd.W0 = upd->W0; f.upd.ld = ldh; f.upd.rows = (int)V;
        f.upd.lr = upd->lr; f.upd.l1 = upd->lambda_1; f.upd.l2 = upd->lambda_2; f.upd.wc = upd->weightcost;
        f.upd.mu = upd->momentum; f.upd.inv_bs = 1.0f / upd->batch_size;
        f.upd.Wp = reinterpret_cast<unsigned short*>(upd->W_planes); f.upd.wp_stride = V * ldh;
        bu.hb = upd->hbias; bu.hbs = upd->hbias_speed; bu.vb = upd->vbias; bu.vbs = upd->vbias_speed;
        bu.H = H; bu.V = V; bu.lr = upd->lr; bu.mu = upd->momentum; bu.inv_rows = 1.0f / upd->n_rows;
        bu.cost_scale = upd->cost_scale; bu.cost_out = upd->cost_out;
    }
    f.fin = make_fin_args(ws.colPpos, ws.colPneg, ws.colV, nb, ldh, ldv, ws.cost_partials, nb * SM_NW, s_h, s_v, cost, fuse_upd ? &bu : nullptr);
    HIP_OK(launch_small_finish(f, s));
    if (upd && !fuse_upd) {
        mdbn_update_args u = *upd;
        u.phase = 0;
        HIP_OK(launch_update(u, s, nullptr, 1, 0, reinterpret_cast<unsigned short*>(u.W_planes)));
    }
    return MDBN_OK;
}

// ---------------------------------------------------------------------------------- thin-batch step (B <= 32, mdbn_thin.hip)
static bool thin_eligible(const mdbn_ctx* ctx, const mdbn_cd_args* a, const Workspace& ws, ThinGeom& tg)
{
    if (!g_opt_thin_fused || g_opt_bf16_inputs) return false;
    if (a->persistent || a->sample_stats || (a->gauss && a->add_noise)) return false;
    if (!a->gauss && a->vs == nullptr) return false;
    if (!thin_geom(a->B, a->V, a->H, a->ldv, a->ldh, std::min(ctx->num_cu, kTargetJobs), tg)) return false;      // (the workspace is sized for <= 256 CUs)
    return (int64_t)tg.G * tg.Bq * a->ldh <= ws.slab_floats && tg.G <= ws.cost_floats;
}

// mode 0: the whole step; 1: gather + positive phase + chain (mdbn_cd_forward); 2: statistics (+ update) (mdbn_cd_statistics).
// upd: single-device step, the update kernel consumes each row of S as it forms it; else it stores [S | s_h | s_v | cost].
static int cd_step_thin(mdbn_ctx* ctx, hipStream_t s, const mdbn_cd_args* a, const mdbn_update_args* upd, const Workspace& ws,
                        const ThinGeom& tg, int mode, const mdbn_update_args* defer)
{
    const int64_t B = a->B, V = a->V, H = a->H, ldv = a->ldv, ldh = a->ldh;
    float* v0 = a->V2;
    float* nv = a->V2 + B * ldv;
    float* ph = a->P2;
    float* nh = a->P2 + B * ldh;
    int n_cost = tg.G;
    // positive phase ahead (mdbn_cd_args.next_indexes on the thin path): the previous call's update kernel left v0 in V2 and
    // the partials of x W in planes_alt; this call's update kernel may do the same for the next one
    const bool keep = a->keep_f32 != 0 || a->trace_h != nullptr || a->trace_v != nullptr;
    const bool ahead_ok = a->planes_alt != nullptr && !keep && tg.lds_ahead > 0 && a->indexes != nullptr;
    const bool ahead_in = ahead_ok && a->v0_ready && mode != 2;
    if (a->planes_alt) REQUIRE(aligned16(a->planes_alt), "planes_alt not 16-byte aligned");
    float* part_ahead = reinterpret_cast<float*>(a->planes_alt);
    if (mode != 2) {
        ThinPassArgs p{};
        p.B = (int)B; p.Bq = tg.Bq; p.V = (int)V; p.H = (int)H; p.ldv = ldv; p.ldh = ldh;
        p.G = tg.G; p.rpw = tg.rpw; p.PW = tg.PW;
        p.W = a->W; p.part = ws.slabs;
        p.data = a->data; p.n_data = a->n_data; p.ld_data = ldv; p.idx = a->indexes; p.idx64 = a->index_is_64; p.v0_out = v0;
        p.vbias = a->vbias; p.gauss = a->gauss;
        p.stamps = g_stamps;
        // x = train_set_x[indexes] and the partials of x W                      (dbn.py:307, rbm.py:303)
        if (!ahead_in) HIP_OK(launch_thin_pass(0, p, tg, s));
        ThinActArgs act{};
        act.part = ahead_in ? part_ahead : ws.slabs; act.G = tg.G; act.Bq = tg.Bq;
        act.e.rows = (int)B; act.e.cols = (int)H; act.e.ld = ldh; act.e.bias = a->hbias;
        act.e.mean = ph; act.e.mean_scale = 1.0f; act.e.sample = a->hs; act.e.rng = make_key(a->rng, 0u);
        HIP_OK(launch_thin_act(act, s));
        act.part = ws.slabs;
        if (a->trace_h) HIP_OK(hipMemcpyAsync(a->trace_h, a->hs, sizeof(float) * B * ldh, hipMemcpyDeviceToDevice, s));
        for (int t = 1; t <= a->k; ++t) {                                  // gibbs_hvh x k (rbm.py:318-336)
            const bool last = t == a->k;
            p.chain = a->hs; p.nv = nv; p.vs = a->gauss ? nullptr : a->vs; p.last = last ? 1 : 0;
            p.target = last ? v0 : nullptr; p.ld_target = ldv; p.cost_partials = last ? ws.cost_partials : nullptr;
            p.rng = make_key(a->rng, (uint32_t)(2 * t - 1));
            HIP_OK(launch_thin_pass(1, p, tg, s));
            if (a->trace_v && !a->gauss)
                HIP_OK(hipMemcpyAsync(a->trace_v + (int64_t)(t - 1) * B * ldv, a->vs, sizeof(float) * B * ldv, hipMemcpyDeviceToDevice, s));
            act.e.mean = nh; act.e.mean_scale = -1.0f; act.e.sample = last ? nullptr : a->hs;
            act.e.rng = make_key(a->rng, (uint32_t)(2 * t));
            HIP_OK(launch_thin_act(act, s));
            if (a->trace_h && !last)
                HIP_OK(hipMemcpyAsync(a->trace_h + (int64_t)t * B * ldh, a->hs, sizeof(float) * B * ldh, hipMemcpyDeviceToDevice, s));
        }
    }
    if (mode == 1) {
        ctx->pending_n_cost = n_cost; ctx->pending_stats = a->stats;
        return MDBN_OK;
    }
    if (mode == 2) n_cost = ctx->pending_n_cost;
    ctx->pending_n_cost = -1; ctx->pending_stats = nullptr;
    // the previous step's deferred update (data-parallel order) is its own launch here, ahead of the statistics
    if (defer) HIP_OK(launch_update(*defer, s, nullptr, 1, 0, reinterpret_cast<unsigned short*>(defer->W_planes)));
    ThinUpdArgs u{};
    u.B = (int)B; u.Bq = tg.Bq; u.V = (int)V; u.H = (int)H; u.ldv = ldv; u.ldh = ldh; u.G = tg.Gu; u.rpw = tg.rpu;
    u.V2 = a->V2; u.P2 = a->P2;
    u.S = a->stats; u.s_h = a->stats + V * ldh; u.s_v = u.s_h + ldh; u.cost = u.s_v + ldv;
    u.cost_partials = ws.cost_partials; u.n_cost = n_cost;
    const bool fuse_upd = upd && g_opt_fused_update;
    u.do_upd = fuse_upd ? 1 : 0;
    if (fuse_upd) {
        u.upd.W = upd->W; u.upd.Ws = upd->W_speed; u.upd.W0 = upd->W0; u.upd.ld = ldh; u.upd.rows = (int)V;
        u.upd.lr = upd->lr; u.upd.l1 = upd->lambda_1; u.upd.l2 = upd->lambda_2; u.upd.wc = upd->weightcost;
        u.upd.mu = upd->momentum; u.upd.inv_bs = 1.0f / upd->batch_size;
        u.upd.Wp = reinterpret_cast<unsigned short*>(upd->W_planes); u.upd.wp_stride = V * ldh;
        u.bu.hb = upd->hbias; u.bu.hbs = upd->hbias_speed; u.bu.vb = upd->vbias; u.bu.vbs = upd->vbias_speed;
        u.bu.H = H; u.bu.V = V; u.bu.lr = upd->lr; u.bu.mu = upd->momentum; u.bu.inv_rows = 1.0f / upd->n_rows;
        u.bu.cost_scale = upd->cost_scale; u.bu.cost_out = upd->cost_out;
    }
    if (fuse_upd && mode == 0 && ahead_ok && a->next_indexes && g_opt_gather_ahead) {
        // update(t) + gather and positive-phase partials of step t + 1 in one pass over W
        u.G = tg.G; u.rpw = tg.rpw; u.PW = tg.PW;
        u.data = a->data; u.n_data = a->n_data; u.ld_data = ldv; u.next_idx = a->next_indexes; u.idx64 = a->index_is_64;
        u.part_next = part_ahead; u.v0_next = a->V2;
        HIP_OK(launch_thin_update_ahead(u, tg, s));
        if (a->ahead_done) *a->ahead_done = 1;
        return MDBN_OK;
    }
    HIP_OK(launch_thin_update(u, tg, s));
    if (upd && !fuse_upd) {
        mdbn_update_args uu = *upd;
        uu.phase = 0;
        HIP_OK(launch_update(uu, s, nullptr, 1, 0, reinterpret_cast<unsigned short*>(uu.W_planes)));
    }
    return MDBN_OK;
}

static int cd_step_impl(mdbn_ctx* ctx, void* stream, const mdbn_cd_args* a, const mdbn_update_args* upd, int mode = 0,
                        const mdbn_update_args* defer = nullptr)
{
    REQUIRE(ctx != nullptr && a != nullptr, "NULL argument");
    REQUIRE(a->struct_size == sizeof(mdbn_cd_args),
            "mdbn_cd_args.struct_size is %llu, this library's struct has %llu bytes (built against another mdbn_hip.h?)",
            (unsigned long long)a->struct_size, (unsigned long long)sizeof(mdbn_cd_args));
    const int64_t B = a->B, V = a->V, H = a->H, ldv = a->ldv, ldh = a->ldh;
    REQUIRE(B > 0 && V > 0 && H > 0 && a->k >= 1, "bad shape / k");
    REQUIRE(a->comm_cus >= 0 && a->comm_cus <= 192, "comm_cus must be in [0, 192]");
    REQUIRE(ldv >= V && ldh >= H && ldv % 4 == 0 && ldh % 4 == 0, "leading dims must be multiples of 4, >= V / H");
    CHECK(check_mat(a->data, ldv, V, "data"));
    CHECK(check_mat(a->W, ldh, H, "W"));
    CHECK(check_mat(a->V2, ldv, V, "V2"));
    CHECK(check_mat(a->P2, ldh, H, "P2"));
    CHECK(check_mat(a->hs, ldh, H, "hs"));
    REQUIRE(a->hbias && a->vbias, "bias pointers are NULL");
    REQUIRE(a->stats != nullptr && aligned16(a->stats), "stats not aligned");
    REQUIRE(a->gauss || a->vs != nullptr, "Bernoulli RBM needs the vs buffer");
    REQUIRE(!(a->sample_stats && a->gauss && a->add_noise), "sample statistics of a noisy GRBM are not supported");
    REQUIRE(a->vs == nullptr || aligned16(a->vs), "vs not aligned");
    REQUIRE(a->persistent == nullptr || aligned16(a->persistent), "persistent not aligned");
    REQUIRE(a->indexes != nullptr || B <= a->n_data, "identity minibatch longer than data");
    Workspace ws;
    CHECK(carve(a->workspace, a->workspace_bytes, B, V, H, ws, true));
    hipStream_t s = (hipStream_t)stream;
    if (a->ahead_done) *a->ahead_done = 0;
    if (upd) {
        REQUIRE(upd->stats == a->stats && upd->W == a->W && upd->ldh == ldh && upd->ldv == ldv && upd->V == V && upd->H == H,
                "update arguments do not match the step's buffers");
        REQUIRE(upd->W_planes == nullptr || upd->W_planes == a->W_planes, "update and step disagree about W_planes");
    }
    // Stale W planes are re-split HERE, whichever path the step then takes: a caller that handed planes in may mark them
    // valid after ANY step (the f32-operand path would otherwise leave stale planes behind a "valid" flag).
    if (a->W_planes && !a->W_planes_valid) {
        REQUIRE(aligned16(a->W_planes), "W_planes not 16-byte aligned");
        HIP_OK(launch_split_planes(a->W, V, ldh, reinterpret_cast<unsigned short*>(a->W_planes), V * ldh, s));
    }
    if (mode == 2)
        REQUIRE(ctx->pending_stats == a->stats && ctx->pending_n_cost >= 0, "mdbn_cd_statistics must follow mdbn_cd_forward of the same step");
    // a whole step between a forward half and its statistics half overwrites the partials that half left in the workspace
    // (shared by all shapes): it ends the hand-over, whichever path serves it
    if (mode == 0) { ctx->pending_n_cost = -1; ctx->pending_stats = nullptr; }
    if (defer) {
        REQUIRE(upd == nullptr && mode != 1, "a deferred update goes with the statistics half of a step without its own update");
        REQUIRE(defer->W == a->W && defer->ldh == ldh && defer->ldv == ldv && defer->V == V && defer->H == H,
                "deferred update does not match the step's parameters");
    }
    if (small_eligible(a, ws) && !(upd && g_opt_update_overlap)) return cd_step_small(ctx, s, a, upd, ws, mode, defer);
    {
        ThinGeom tg;
        if (thin_eligible(ctx, a, ws, tg) && !(upd && g_opt_update_overlap)) return cd_step_thin(ctx, s, a, upd, ws, tg, mode, defer);
    }
    if (planes_eligible(a) && !(upd && g_opt_update_overlap)) {
        const int rc = cd_step_planes(ctx, s, a, upd, ws, mode, defer);
        if (mode != 1) { ctx->pending_n_cost = -1; ctx->pending_stats = nullptr; }
        return rc;
    }

    float* v0 = a->V2;
    float* nv = a->V2 + B * ldv;
    float* ph = a->P2;
    float* nh = a->P2 + B * ldh;
    int n_cost = 0;
    // Mid-size layers: gather + positive phase + the whole Gibbs chain in ONE launch on groups of workgroups that hold W in
    // their LDS between them (mdbn_gchain.hip); it leaves V2 / P2 / the column and cost partials exactly as the launches
    // below would, so the statistics half of the step is unchanged
    GChainGeom gg;
    const bool use_gchain = mode != 2 && g_opt_gchain && !g_opt_bf16_inputs && !a->persistent && !a->sample_stats &&
                            !(a->gauss && a->add_noise) && (a->gauss || a->vs != nullptr) &&
                            gchain_geom(B, V, H, ldv, ldh, a->gauss, std::min(ctx->num_cu, kTargetJobs), gg) &&
                            gg.xbuf_floats <= ws.slab_floats && (int64_t)gg.nslab * gg.g <= ws.cost_floats;
    if (use_gchain) {
        if (!ctx->gc_flags) {
            HIP_OK(hipMalloc(reinterpret_cast<void**>(&ctx->gc_flags), sizeof(unsigned) * (GC_MAX_FLAGS + 4)));
            HIP_OK(hipMemsetAsync(ctx->gc_flags, 0, sizeof(unsigned) * (GC_MAX_FLAGS + 4), s));
        }
        GChainArgs c{};
        c.B = (int)B; c.V = (int)V; c.H = (int)H; c.k = a->k; c.gauss = a->gauss; c.ldv = ldv; c.ldh = ldh;
        c.g = gg.g; c.Vb = gg.Vb; c.nslab = gg.nslab; c.nsg = gg.nsg; c.PW = gg.PW; c.S1 = gg.S1;
        c.W = a->W; c.hbias = a->hbias; c.vbias = a->vbias;
        c.data = a->data; c.n_data = a->n_data; c.ld_data = ldv; c.idx = a->indexes; c.idx64 = a->index_is_64;
        c.V2 = a->V2; c.P2 = a->P2; c.hs = a->hs; c.vs = a->gauss ? nullptr : a->vs;
        c.trace_h = a->trace_h; c.trace_v = a->gauss ? nullptr : a->trace_v;
        c.colPpos = ws.colPpos; c.colPneg = ws.colPneg; c.colV = ws.colV; c.cost_partials = ws.cost_partials;
        c.xbuf = ws.slabs; c.flags = ctx->gc_flags; c.error = ctx->gc_flags + GC_MAX_FLAGS;
        c.seq0 = ctx->gc_seq;
        ctx->gc_seq += (unsigned)(((gg.nslab + gg.nsg - 1) / gg.nsg) * (a->k + 1)) + 1u;
        c.rng = make_key(a->rng, 0u);
        HIP_OK(launch_gchain(c, gg.lds, s));
        n_cost = gg.nslab * gg.g;
    }
    if (mode != 2 && !use_gchain) {

    // x = train_set_x[indexes]                                        (dbn.py:307)
    // (its own launch: read through the index list inside the first propup of the streaming kernel, the rows come from HBM
    //  at HBM latency into every workgroup's operand stream -- bit-identical and 3 - 10 us per step SLOWER,
    //  profiles/r05zi_stream_gather_ab.log; gathered AHEAD by extra workgroups of the previous step's statistics launch
    //  into a second V2 buffer: bit-identical too, and that launch grows by more than the gather launch it saves,
    //  profiles/r05zm_gather_ahead_dense_ab.log, r05zl_*)
    HIP_OK(launch_gather(a->data, a->n_data, ldv, ldv, a->indexes, a->index_is_64, B, v0, ldv, s));

    // positive phase: ph_mean, ph_sample                              (rbm.py:303)
    {
        Affine up{v0, B, ldv, a->W, V, H, ldh, 0, a->hbias, nullptr, ph, a->hs, ldh, 1.0f, 0,
                  nullptr, 0, false, &a->rng, 0u};
        up.colsum = ws.colPpos;                                         // sum_rows ph_mean
        CHECK(run_affine(up, ws, s, nullptr));
        if (a->trace_h) HIP_OK(hipMemcpyAsync(a->trace_h, a->hs, sizeof(float) * B * ldh, hipMemcpyDeviceToDevice, s));
    }
    for (int t = 1; t <= a->k; ++t) {                                  // gibbs_hvh x k (rbm.py:318-336)
        const bool last = t == a->k;
        const float* chain = (t == 1 && a->persistent) ? a->persistent : a->hs;   // rbm.py:308-311
        // v_t | h_{t-1}: RBM sigmoid + Bernoulli (rbm.py:229-240); GRBM linear mean (rbm.py:647-660;
        // its noisy sample never feeds the chain, rbm.py:669, so it is not materialised here)
        // sample_stats (compute_symbolic_grad, rbm.py:339-342,378-390): the negative visible data is
        // the SAMPLE nv_samples[-1]; it then also is the input of the last propup, as in rbm.py:246
        const bool samp_stats = last && a->sample_stats && !a->gauss;
        Affine down{chain, B, ldh, a->W, V, H, ldh, 1, a->vbias, nullptr, samp_stats ? nullptr : nv,
                    a->gauss ? nullptr : (samp_stats ? nv : a->vs),
                    ldv, 1.0f, a->gauss, last ? v0 : nullptr, ldv, last, &a->rng, (uint32_t)(2 * t - 1)};
        down.x_binary = chain == a->hs;         // our own 0/1 hidden samples (a caller's persistent chain may hold anything)
        if (last) { down.colsum = ws.colV; down.colsum_kind = samp_stats ? 2 : 1; }   // sum_rows (v0 - nv)
        CHECK(run_affine(down, ws, s, last ? &n_cost : nullptr));
        if (a->trace_v && !a->gauss)
            HIP_OK(hipMemcpyAsync(a->trace_v + (int64_t)(t - 1) * B * ldv, samp_stats ? nv : a->vs, sizeof(float) * B * ldv,
                                  hipMemcpyDeviceToDevice, s));
        // h_t | v_t: from the mean for GRBM (rbm.py:669), from the sample for RBM (rbm.py:246)
        const bool need_sample = !last || a->persistent != nullptr;
        float* hdst = (last && a->persistent) ? a->persistent : a->hs;            // rbm.py:369
        Affine up{(a->gauss || samp_stats) ? nv : a->vs, B, ldv, a->W, V, H, ldh, 0, a->hbias, nullptr, nh,
                  need_sample ? hdst : nullptr, ldh, -1.0f, 0, nullptr, 0, false, &a->rng, (uint32_t)(2 * t)};
        up.x_binary = !a->gauss;                // Bernoulli visibles: the chain feeds the 0/1 sample upward
        if (last) up.colsum = ws.colPneg;                               // sum_rows (-nh_mean)
        CHECK(run_affine(up, ws, s, nullptr));
        if (a->trace_h && need_sample)
            HIP_OK(hipMemcpyAsync(a->trace_h + (int64_t)t * B * ldh, hdst, sizeof(float) * B * ldh, hipMemcpyDeviceToDevice, s));
    }

    }       // mode != 2
    if (mode == 1) {
        ctx->pending_n_cost = n_cost; ctx->pending_stats = a->stats;
        return MDBN_OK;
    }
    if (mode == 2) { n_cost = ctx->pending_n_cost; ctx->pending_n_cost = -1; ctx->pending_stats = nullptr; }
    // (f32-operand kernels: the previous step's deferred update is its own launch, ahead of the statistics GEMM)
    if (defer) HIP_OK(launch_update(*defer, s, nullptr, 1, 0, reinterpret_cast<unsigned short*>(defer->W_planes)));

    float* S = a->stats;
    float* s_h = a->stats + V * ldh;
    float* s_v = s_h + ldh;
    float* cost = s_v + ldv;

    // The statistics GEMM reads only V2 / P2 -- never W -- and the parameter half of the update
    // (theta * m + OLD speed * lr, rbm.py:364-365) does not need its result.  With an update
    // attached (single device) the bias/cost finalize and that parameter half run on a side
    // stream UNDER the compute-bound GEMM; only the speed half waits for S.
    bool overlap = false;
    mdbn_update_args u;
    if (upd) {
        u = *upd;
        REQUIRE(u.stats == a->stats && u.W == a->W && u.ldh == ldh && u.ldv == ldv && u.V == V && u.H == H,
                "update arguments do not match the step's buffers");
        overlap = g_opt_update_overlap != 0 && u.lambda_1 == 0.f && (u.weightcost == 0.f || u.W0 != nullptr);
        if (overlap && ctx->side == nullptr) {
            HIP_OK(hipStreamCreateWithFlags(&ctx->side, hipStreamNonBlocking));
            HIP_OK(hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
            HIP_OK(hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming));
        }
    }
    if (overlap) {
        HIP_OK(hipEventRecord(ctx->ev_fork, s));
        HIP_OK(hipStreamWaitEvent(ctx->side, ctx->ev_fork, 0));
        HIP_OK(launch_finalize_stats(ws.colPpos, ws.colPneg, ws.colV, row_groups(B), ldh, ldv, ws.cost_partials,
                                     n_cost, s_h, s_v, cost, nullptr, ctx->side));
        u.phase = 2;
        HIP_OK(launch_update(u, ctx->side, nullptr, 1, 0, reinterpret_cast<unsigned short*>(u.W_planes)));
        HIP_OK(hipEventRecord(ctx->ev_join, ctx->side));
    }
    const Plan p = plan_stats(V, H, 2 * B, ldh);
    // Single device, unsplit statistics GEMM: the GEMM applies the weight update to its own tiles
    // (S never reaches HBM) and the bias / cost half rides on the finalize kernel -- no update launch.
    const bool fuse_upd = upd && !overlap && g_opt_fused_update && p.splitk == 1;
    // ... and in the LDS-tiled kernel even the finalize units run inside the GEMM (its MFMA waves are
    // idle while the first slice is in flight): no finalize launch either
    const bool fin_in_gemm = fuse_upd && g_opt_fused_finalize;
    BiasUpd bu;
    if (fuse_upd) {
        bu.hb = u.hbias; bu.hbs = u.hbias_speed; bu.vb = u.vbias; bu.vbs = u.vbias_speed;
        bu.H = H; bu.V = V; bu.lr = u.lr; bu.mu = u.momentum; bu.inv_rows = 1.0f / u.n_rows;
        bu.cost_scale = u.cost_scale; bu.cost_out = u.cost_out;
    }
    // the unfused LDS-tiled statistics GEMM (data-parallel step) runs the plain finalize units the same way
    const bool fin_in_plain_gemm = !fuse_upd && !overlap && g_opt_fused_finalize;
    if (!overlap && !fin_in_gemm && !fin_in_plain_gemm)
        HIP_OK(launch_finalize_stats(ws.colPpos, ws.colPneg, ws.colV, row_groups(B), ldh, ldv, ws.cost_partials,
                                     n_cost, s_h, s_v, cost, fuse_upd ? &bu : nullptr, s));
    GemmArgs g{};
    g.A = a->V2; g.lda = ldv; g.B = a->P2; g.ldb = ldh;
    g.ldc = ldh; g.slab_stride = V * ldh;
    g.M = (int)V; g.N = (int)H; g.K = (int)(2 * B); g.Nst = (int)ldh;
    p.fill(g);
    g.fin_enabled = 0;
    if (fuse_upd) {
        g.C = nullptr;
        g.fused = 2;
        if (fin_in_gemm) {
            g.fin_enabled = 1;
            g.fin = make_fin_args(ws.colPpos, ws.colPneg, ws.colV, row_groups(B), ldh, ldv, ws.cost_partials, n_cost,
                                  s_h, s_v, cost, &bu);
        }
        g.upd.W = u.W; g.upd.Ws = u.W_speed; g.upd.W0 = u.W0; g.upd.ld = ldh; g.upd.rows = (int)V;
        g.upd.lr = u.lr; g.upd.l1 = u.lambda_1; g.upd.l2 = u.lambda_2; g.upd.wc = u.weightcost;
        g.upd.mu = u.momentum; g.upd.inv_bs = 1.0f / u.batch_size;
        g.upd.Wp = reinterpret_cast<unsigned short*>(u.W_planes); g.upd.wp_stride = V * ldh;
        HIP_OK(timed_gemm(LAY_MN, LAY_MN, g, s));
        return MDBN_OK;
    }
    if (fin_in_plain_gemm) {
        g.fin_enabled = 1;
        g.fin = make_fin_args(ws.colPpos, ws.colPneg, ws.colV, row_groups(B), ldh, ldv, ws.cost_partials, n_cost,
                              s_h, s_v, cost, nullptr);
    }
    if (p.splitk == 1) {
        g.C = S;
        HIP_OK(timed_gemm(LAY_MN, LAY_MN, g, s));
    } else {
        g.C = ws.slabs;
        HIP_OK(timed_gemm(LAY_MN, LAY_MN, g, s));
        if (upd && !overlap && g_opt_fused_update) {
            // single device: the update sums the slabs itself (same order as sum_slabs_kernel, so the
            // same bits); S is not materialised, as in the fused unsplit case
            u.phase = 0;
            HIP_OK(launch_update(u, s, ws.slabs, p.splitk, g.slab_stride, reinterpret_cast<unsigned short*>(u.W_planes)));
            return MDBN_OK;
        }
        HIP_OK(launch_sum_slabs(ws.slabs, p.splitk, g.slab_stride, V * ldh, S, s));
    }
    if (upd) {
        if (overlap) {
            HIP_OK(hipStreamWaitEvent(s, ctx->ev_join, 0));
            u.phase = 1;
        } else {
            u.phase = 0;
        }
        HIP_OK(launch_update(u, s, nullptr, 1, 0, u.phase != 1 ? reinterpret_cast<unsigned short*>(u.W_planes) : nullptr));
    }
    return MDBN_OK;
}

int mdbn_cd_step(mdbn_ctx* ctx, void* stream, const mdbn_cd_args* a)
{
    CtxScope ctx_scope(ctx);
    return cd_step_impl(ctx, stream, a, nullptr);
}

int mdbn_cd_forward(mdbn_ctx* ctx, void* stream, const mdbn_cd_args* a)
{
    CtxScope ctx_scope(ctx);
    return cd_step_impl(ctx, stream, a, nullptr, 1, nullptr);
}

int mdbn_cd_statistics(mdbn_ctx* ctx, void* stream, const mdbn_cd_args* a, const mdbn_update_args* deferred)
{
    CtxScope ctx_scope(ctx);
    if (deferred) {
        CHECK(check_update_args(deferred));
        REQUIRE(deferred->phase == 3 || deferred->phase == 0, "a deferred update is phase 3 (or the whole rule, phase 0)");
    }
    return cd_step_impl(ctx, stream, a, nullptr, 2, deferred);
}

int mdbn_cd_train_step(mdbn_ctx* ctx, void* stream, const mdbn_cd_args* a, const mdbn_update_args* upd)
{
    CtxScope ctx_scope(ctx);
    REQUIRE(upd != nullptr, "update arguments are NULL");
    CHECK(check_update_args(upd));
    return cd_step_impl(ctx, stream, a, upd);
}

int mdbn_free_energy(mdbn_ctx* ctx, void* stream, const float* x, int64_t N, int64_t ldv, const float* W,
                     int64_t V, int64_t H, int64_t ldh, const float* hbias, const float* vbias, int gauss,
                     float* out, void* workspace, int64_t workspace_bytes)
{
    CtxScope ctx_scope(ctx);
    REQUIRE(ctx != nullptr, "ctx is NULL");
    REQUIRE(N > 0 && V > 0 && H > 0, "bad shape");
    CHECK(check_mat(x, ldv, V, "x"));
    CHECK(check_mat(W, ldh, H, "W"));
    REQUIRE(hbias && vbias && out, "NULL pointer");
    Workspace ws;
    CHECK(carve(workspace, workspace_bytes, N, V, H, ws, false));
    hipStream_t s = (hipStream_t)stream;
    int64_t r0 = 0;
    while (r0 < N) {
        int64_t R = N - r0;
        Plan p = plan_gemm(R, H, V);
        while (p.slab_floats(R, ldh) > ws.slab_floats) {
            if (R <= 4) return fail(MDBN_ENOSPC, "workspace cannot hold one 4-row slab");
            R = std::max<int64_t>(4, (R / 2 + 3) & ~int64_t(3));
            p = plan_gemm(R, H, V);
        }
        GemmArgs g{};
        g.A = x + r0 * ldv; g.lda = ldv; g.B = W; g.ldb = ldh; g.C = ws.slabs; g.ldc = ldh;
        g.slab_stride = R * ldh;
        g.M = (int)R; g.N = (int)H; g.K = (int)V; g.Nst = (int)ldh;
        p.fill(g);
        g.fused = 0;
        HIP_OK(timed_gemm(LAY_K, LAY_MN, g, s));
        HIP_OK(launch_free_energy(ws.slabs, p.splitk, g.slab_stride, ldh, (int)H, hbias, x + r0 * ldv, ldv,
                                  (int)V, vbias, gauss, R, out + r0, s));
        r0 += R;
    }
    return MDBN_OK;
}

int mdbn_round_flip(mdbn_ctx* ctx, void* stream, const float* x, int64_t rows, int64_t cols, int64_t ld, int64_t flip_col,
                    float* out)
{
    CtxScope ctx_scope(ctx);
    REQUIRE(ctx && x && out && rows >= 0 && cols > 0 && ld >= cols && flip_col < cols, "bad arguments");
    HIP_OK(launch_round_flip(x, out, rows, cols, ld, flip_col, (hipStream_t)stream));
    return MDBN_OK;
}

int mdbn_pl_cost(mdbn_ctx* ctx, void* stream, const float* fe, const float* fe_flip, int64_t rows, int64_t n_visible,
                 float* cost_out)
{
    CtxScope ctx_scope(ctx);
    REQUIRE(ctx && fe && fe_flip && cost_out && rows > 0 && n_visible > 0, "bad arguments");
    HIP_OK(launch_pl_cost(fe, fe_flip, rows, (float)n_visible, cost_out, (hipStream_t)stream));
    return MDBN_OK;
}

int mdbn_recon_cost(mdbn_ctx* ctx, void* stream, const float* pre, int64_t ld_pre, const float* target, int64_t ld_target,
                    int64_t rows, int64_t cols, int gauss, float* cost_out, void* workspace, int64_t workspace_bytes)
{
    CtxScope ctx_scope(ctx);
    REQUIRE(ctx && pre && target && cost_out && rows > 0 && cols > 0 && ld_pre >= cols && ld_target >= cols, "bad arguments");
    REQUIRE(workspace != nullptr && workspace_bytes >= 4096, "workspace of >= 4096 bytes needed");
    const int nb = (int)std::min<int64_t>(std::min<int64_t>(1024, workspace_bytes / 4), (rows * cols + 255) / 256);
    // rbm.py:479-480: sum over units, mean over rows; rbm.py:697: mean over everything
    const float scale = gauss ? 1.0f / ((float)rows * (float)cols) : 1.0f / (float)rows;
    HIP_OK(launch_recon_cost(pre, ld_pre, target, ld_target, rows, cols, gauss, scale, reinterpret_cast<float*>(workspace), nb,
                             cost_out, (hipStream_t)stream));
    return MDBN_OK;
}

int mdbn_tanh(mdbn_ctx* ctx, void* stream, float* x, int64_t rows, int64_t cols, int64_t ld)
{
    CtxScope ctx_scope(ctx);
    REQUIRE(ctx && x && rows >= 0 && cols > 0 && ld >= cols, "bad arguments");
    HIP_OK(launch_tanh(x, rows, cols, ld, (hipStream_t)stream));
    return MDBN_OK;
}

int mdbn_count_nonfinite(mdbn_ctx* ctx, void* stream, const float* x, int64_t n, int32_t* count)
{
    CtxScope ctx_scope(ctx);
    REQUIRE(ctx && x && count && n >= 0, "bad arguments");
    HIP_OK(launch_count_nonfinite(x, n, count, (hipStream_t)stream));
    return MDBN_OK;
}

int mdbn_f32_to_bf16(mdbn_ctx* ctx, void* stream, const float* src, void* dst, int64_t n)
{
    CtxScope ctx_scope(ctx);
    REQUIRE(ctx && src && dst && n >= 0, "bad arguments");
    REQUIRE(aligned16(src) && aligned16(dst), "buffers must be 16-byte aligned");
    HIP_OK(launch_narrow_bf16(src, reinterpret_cast<unsigned short*>(dst), n, (hipStream_t)stream));
    return MDBN_OK;
}

int mdbn_bf16_to_f32(mdbn_ctx* ctx, void* stream, const void* src, float* dst, int64_t n)
{
    CtxScope ctx_scope(ctx);
    REQUIRE(ctx && src && dst && n >= 0, "bad arguments");
    REQUIRE(aligned16(src) && aligned16(dst), "buffers must be 16-byte aligned");
    HIP_OK(launch_widen_bf16(reinterpret_cast<const unsigned short*>(src), dst, n, (hipStream_t)stream));
    return MDBN_OK;
}

int mdbn_rng_uniform(mdbn_ctx* ctx, void* stream, float* out, int64_t rows, int64_t cols, int64_t ld,
                     const mdbn_rng* rng)
{
    CtxScope ctx_scope(ctx);
    REQUIRE(ctx && out && rng && rows >= 0 && cols >= 0 && ld >= cols, "bad arguments");
    HIP_OK(launch_rng_fill(out, rows, cols, ld, make_key(*rng, rng->draw), 0, (hipStream_t)stream));
    return MDBN_OK;
}

int mdbn_rng_normal(mdbn_ctx* ctx, void* stream, float* out, int64_t rows, int64_t cols, int64_t ld,
                    const mdbn_rng* rng)
{
    CtxScope ctx_scope(ctx);
    REQUIRE(ctx && out && rng && rows >= 0 && cols >= 0 && ld >= cols, "bad arguments");
    HIP_OK(launch_rng_fill(out, rows, cols, ld, make_key(*rng, rng->draw), 1, (hipStream_t)stream));
    return MDBN_OK;
}

int mdbn_philox_host(float* out, int64_t rows, int64_t cols, int64_t ld, const mdbn_rng* rng)
{
    REQUIRE(out && rng && rows >= 0 && cols >= 0 && ld >= cols, "bad arguments");
    const PhiloxKey k = make_key(*rng, rng->draw);
    for (int64_t r = 0; r < rows; ++r) {
        const uint64_t g = k.row_offset + (uint64_t)r;
        for (int64_t c = 0; c < cols; ++c) {
            uint32_t w[4];
            philox4x32_10((uint32_t)c, (uint32_t)(g >> 2), k.draw, k.step, k.k0, k.k1, w);
            out[r * ld + c] = philox_u01(w[g & 3]);
        }
    }
    return MDBN_OK;
}

// ---------------------------------------------------------------------------------- row feeder (host-resident table)
// Minibatch rows of a table that stays in host memory reach the device WITHOUT a kernel on the CUs: worker threads
// gather the rows into a pinned staging slot, one hipMemcpyAsync per minibatch moves the slot on the feeder's own copy
// stream (SDMA), the consuming stream waits for the copy's event.  A kernel reading the pinned table over PCIe beside the
// step slows every GEMM of the step (one-workgroup-per-CU grids: step 145 -> 215 us); an SDMA copy
// does not (145.7 -> 147.3 us).  The ring is `slots` deep so the copy of minibatch t + 2 runs beside step t.
using mdbn_host::RowPool;

struct mdbn_feeder {
    int device = 0;
    const float* table = nullptr; int64_t n_rows = 0, cols = 0, ld = 0, max_rows = 0, ld_dev = 0;
    int slots = 0;
    std::vector<float*> dev, staging;
    std::vector<hipEvent_t> copied, consumed;
    std::vector<char> copied_set, consumed_set, busy;
    hipStream_t copy_stream = nullptr, copy_stream2 = nullptr;     // the second one only with feed_copy_streams = 2
    hipEvent_t half_done = nullptr, half_go = nullptr;
    RowPool* pool = nullptr;
    std::thread dispatcher;
    std::mutex m;
    std::condition_variable cv;
    struct Job { int64_t ticket; std::vector<int64_t> idx; bool identity; int64_t n; };
    struct Done { int slot; int rc; int64_t n; };
    struct Staged { int64_t ticket; int slot; int rc; int64_t n; };
    std::deque<Job> queue;                        // submitted
    std::deque<Staged> staged;                    // gathered into pinned staging, copy not yet enqueued
    std::set<int64_t> inflight;                   // taken off `queue`, not yet in `ready`
    std::map<int64_t, Done> ready;                // copy enqueued, not yet acquired
    std::map<int64_t, int> held;                  // acquired, not yet released: ticket -> slot
    std::thread copier;
    int64_t next_ticket = 0;
    int next_slot = 0;
    bool stop = false;
    // host-side time spent per stage (mdbn_feeder_stats), seconds
    double t_gather = 0, t_copy_call = 0, t_acquire_wait = 0;
    int64_t n_fed = 0, n_acquired = 0;
    static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

    // stage 1 (dispatcher + pool): rows -> pinned staging slot
    void run()
    {
        for (;;) {
            Job job;
            int slot;
            {
                std::unique_lock<std::mutex> l(m);
                cv.wait(l, [&] { return stop || (!queue.empty() && !busy[next_slot]); });
                if (stop) return;
                job = std::move(queue.front());
                queue.pop_front();
                slot = next_slot;
                next_slot = (next_slot + 1) % slots;
                busy[slot] = 1;
                inflight.insert(job.ticket);
            }
            int rc = MDBN_OK;
            // the staging slot is free once the copy that last read it has finished (long ago in steady state)
            if (copied_set[slot] && hipEventSynchronize(copied[slot]) != hipSuccess) rc = MDBN_EHIP;
            const double t0 = now();
            if (rc == MDBN_OK && !pool->gather(table, n_rows, cols, ld, job.identity ? nullptr : job.idx.data(), job.n,
                                               staging[slot], ld_dev))
                rc = MDBN_EINVAL;
            {
                std::lock_guard<std::mutex> l(m);
                t_gather += now() - t0;
                staged.push_back(Staged{job.ticket, slot, rc, job.n});
            }
            cv.notify_all();
        }
    }
    // stage 2 (copier): staging slot -> device slot, one copy on the copy stream.  Its own thread: enqueueing an 8-MB
    // hipMemcpyAsync takes the host ~100 us, during which stage 1 gathers the next minibatch.
    void copy_loop()
    {
        (void)hipSetDevice(device);
        for (;;) {
            Staged st;
            {
                std::unique_lock<std::mutex> l(m);
                cv.wait(l, [&] { return stop || !staged.empty(); });
                if (stop) return;
                st = staged.front();
                staged.pop_front();
            }
            int rc = st.rc;
            const double t0 = now();
            if (rc == MDBN_OK) {
                // the device slot is free once the step that last read it has finished: stream order, no host wait
                hipError_t e = consumed_set[st.slot] ? hipStreamWaitEvent(copy_stream, consumed[st.slot], 0) : hipSuccess;
                const int64_t n1 = copy_stream2 ? (st.n + 1) / 2 : st.n;       // rows of the first copy
                if (e == hipSuccess && copy_stream2 && st.n > n1) {
                    // second half on the second stream, ordered after the same "slot is free" point
                    e = hipEventRecord(half_go, copy_stream);
                    if (e == hipSuccess) e = hipStreamWaitEvent(copy_stream2, half_go, 0);
                    if (e == hipSuccess)
                        e = hipMemcpyAsync(dev[st.slot] + n1 * ld_dev, staging[st.slot] + n1 * ld_dev,
                                           (size_t)(st.n - n1) * ld_dev * sizeof(float), hipMemcpyHostToDevice, copy_stream2);
                    if (e == hipSuccess) e = hipEventRecord(half_done, copy_stream2);
                }
                if (e == hipSuccess)
                    e = hipMemcpyAsync(dev[st.slot], staging[st.slot], (size_t)n1 * ld_dev * sizeof(float), hipMemcpyHostToDevice,
                                       copy_stream);
                if (e == hipSuccess && copy_stream2 && st.n > n1) e = hipStreamWaitEvent(copy_stream, half_done, 0);
                if (e == hipSuccess) e = hipEventRecord(copied[st.slot], copy_stream);
                if (e != hipSuccess) rc = MDBN_EHIP;
                else copied_set[st.slot] = 1;
            }
            {
                std::lock_guard<std::mutex> l(m);
                ready[st.ticket] = Done{st.slot, rc, st.n};
                inflight.erase(st.ticket);
                t_copy_call += now() - t0;
                ++n_fed;
            }
            cv.notify_all();
        }
    }
};

int mdbn_host_gather_rows(const float* table, int64_t n_rows, int64_t cols, int64_t ld, const int64_t* indexes, int64_t n,
                          float* out, int64_t ld_out, int threads)
{
    REQUIRE(table && out && n_rows > 0 && cols > 0 && ld >= cols && ld_out >= cols && n >= 0, "bad arguments");
    REQUIRE(threads >= 1 && threads <= 64, "threads must be in [1, 64]");
    REQUIRE(indexes != nullptr || n <= n_rows, "identity gather longer than the table");
    RowPool pool(threads - 1);
    if (!pool.gather(table, n_rows, cols, ld, indexes, n, out, ld_out)) return fail(MDBN_EINVAL, "row index out of range");
    return MDBN_OK;
}

int mdbn_feeder_create(mdbn_ctx* ctx, const float* table, int64_t n_rows, int64_t cols, int64_t ld, int64_t max_rows,
                       int slots, float* const* device_slots, int64_t ld_device, int threads, mdbn_feeder** out)
{
    CtxScope ctx_scope(ctx);
    REQUIRE(ctx != nullptr && out != nullptr, "ctx / out is NULL");
    REQUIRE(table && n_rows > 0 && cols > 0 && ld >= cols && max_rows > 0, "bad table");
    REQUIRE(slots >= 2 && slots <= 16 && device_slots != nullptr, "slots must be in [2, 16]");
    REQUIRE(ld_device >= cols && ld_device % 4 == 0, "bad device leading dimension");
    REQUIRE(threads >= 1 && threads <= 64, "threads must be in [1, 64]");
    for (int i = 0; i < slots; ++i) REQUIRE(device_slots[i] && aligned16(device_slots[i]), "device slot NULL / not 16-byte aligned");
    HIP_OK(hipSetDevice(ctx->device));
    mdbn_feeder* f = new mdbn_feeder;
    f->device = ctx->device;
    f->table = table; f->n_rows = n_rows; f->cols = cols; f->ld = ld; f->max_rows = max_rows; f->ld_dev = ld_device;
    f->slots = slots;
    f->copied_set.assign(slots, 0); f->consumed_set.assign(slots, 0); f->busy.assign(slots, 0);
    hipError_t e = hipStreamCreateWithFlags(&f->copy_stream, hipStreamNonBlocking);
    if (e == hipSuccess && g_opt_feed_copy_streams == 2) {
        e = hipStreamCreateWithFlags(&f->copy_stream2, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&f->half_done, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&f->half_go, hipEventDisableTiming);
    }
    for (int i = 0; i < slots && e == hipSuccess; ++i) {
        float* st = nullptr;
        hipEvent_t a = nullptr, b = nullptr;
        e = hipHostMalloc(reinterpret_cast<void**>(&st), (size_t)max_rows * ld_device * sizeof(float), hipHostMallocDefault);
        if (e == hipSuccess) { memset(st, 0, (size_t)max_rows * ld_device * sizeof(float)); f->staging.push_back(st); }   // pad columns stay zero
        if (e == hipSuccess) e = hipEventCreateWithFlags(&a, hipEventDisableTiming);
        if (e == hipSuccess) { f->copied.push_back(a); e = hipEventCreateWithFlags(&b, hipEventDisableTiming); }
        if (e == hipSuccess) f->consumed.push_back(b);
        f->dev.push_back(device_slots[i]);
    }
    if (e != hipSuccess) {
        for (float* st : f->staging) (void)hipHostFree(st);
        for (hipEvent_t ev : f->copied) (void)hipEventDestroy(ev);
        for (hipEvent_t ev : f->consumed) (void)hipEventDestroy(ev);
        if (f->half_done) (void)hipEventDestroy(f->half_done);
        if (f->half_go) (void)hipEventDestroy(f->half_go);
        if (f->copy_stream2) (void)hipStreamDestroy(f->copy_stream2);
        if (f->copy_stream) (void)hipStreamDestroy(f->copy_stream);
        delete f;
        return fail(MDBN_EHIP, "row feeder: %s", hipGetErrorString(e));
    }
    f->pool = new RowPool(threads - 1);           // the dispatcher thread is the pool's last worker
    f->dispatcher = std::thread([f] { f->run(); });
    f->copier = std::thread([f] { f->copy_loop(); });
    *out = f;
    return MDBN_OK;
}

int mdbn_feeder_submit(mdbn_feeder* f, const int64_t* indexes, int64_t n, int64_t* ticket)
{
    REQUIRE(f != nullptr && ticket != nullptr, "feeder / ticket is NULL");
    REQUIRE(n > 0 && n <= f->max_rows, "row count outside (0, max_rows]");
    REQUIRE(indexes != nullptr || n <= f->n_rows, "identity gather longer than the table");
    mdbn_feeder::Job job;
    job.identity = indexes == nullptr;
    job.n = n;
    if (indexes) job.idx.assign(indexes, indexes + n);          // the caller's array may go away
    {
        std::lock_guard<std::mutex> l(f->m);
        job.ticket = *ticket = f->next_ticket++;
        f->queue.push_back(std::move(job));
    }
    f->cv.notify_all();
    return MDBN_OK;
}

int mdbn_feeder_acquire(mdbn_feeder* f, int64_t ticket, void* stream, int* slot)
{
    REQUIRE(f != nullptr && slot != nullptr, "feeder / slot is NULL");
    mdbn_feeder::Done d;
    {
        std::unique_lock<std::mutex> l(f->m);
        bool pending = f->ready.count(ticket) != 0 || f->inflight.count(ticket) != 0;
        if (!pending)
            for (const auto& j : f->queue) pending = pending || j.ticket == ticket;
        if (!pending) return fail(MDBN_EINVAL, "row feeder: ticket %lld is not pending", (long long)ticket);
        // tickets are served in submission order on a ring of `slots`: with every slot held by the caller the ring cannot
        // reach this ticket
        if (!f->ready.count(ticket) && (int)f->held.size() >= f->slots)
            return fail(MDBN_EINVAL, "row feeder: all %d slots are held; release one before acquiring ticket %lld", f->slots,
                        (long long)ticket);
        // ... and with every slot taken by EARLIER tickets that only this caller can free (ready or about to be, or held),
        // a ticket that has not been started yet never will be: refuse instead of blocking for ever (slots = 3, tickets
        // 0..3 submitted, acquire(3))
        if (!f->ready.count(ticket) && !f->inflight.count(ticket) &&
            (int)(f->ready.size() + f->held.size() + f->inflight.size()) >= f->slots)
            return fail(MDBN_EINVAL, "row feeder: ticket %lld cannot start while %d earlier tickets occupy all %d slots; "
                        "acquire and release them first (tickets are served in submission order)", (long long)ticket,
                        (int)(f->ready.size() + f->held.size() + f->inflight.size()), f->slots);
        const double t0 = mdbn_feeder::now();
        f->cv.wait(l, [&] { return f->ready.count(ticket) != 0 || f->stop; });
        f->t_acquire_wait += mdbn_feeder::now() - t0;
        ++f->n_acquired;
        if (!f->ready.count(ticket)) return fail(MDBN_EINVAL, "row feeder: shut down");
        d = f->ready[ticket];
        f->ready.erase(ticket);
        if (d.rc != MDBN_OK) {
            f->busy[d.slot] = 0;
            l.unlock();
            f->cv.notify_all();
            return fail(d.rc, d.rc == MDBN_EINVAL ? "row feeder: row index out of range" : "row feeder: a HIP call failed");
        }
        f->held[ticket] = d.slot;
    }
    HIP_OK(hipStreamWaitEvent((hipStream_t)stream, f->copied[d.slot], 0));
    *slot = d.slot;
    return MDBN_OK;
}

int mdbn_feeder_release(mdbn_feeder* f, int64_t ticket, void* stream)
{
    REQUIRE(f != nullptr, "feeder is NULL");
    int slot;
    {
        std::lock_guard<std::mutex> l(f->m);
        auto it = f->held.find(ticket);
        if (it == f->held.end()) return fail(MDBN_EINVAL, "row feeder: ticket %lld is not held", (long long)ticket);
        slot = it->second;
        f->held.erase(it);
    }
    HIP_OK(hipEventRecord(f->consumed[slot], (hipStream_t)stream));
    {
        std::lock_guard<std::mutex> l(f->m);
        f->consumed_set[slot] = 1;
        f->busy[slot] = 0;
    }
    f->cv.notify_all();
    return MDBN_OK;
}

int mdbn_feeder_stats(mdbn_feeder* f, double* out5)
{
    REQUIRE(f != nullptr && out5 != nullptr, "feeder / out is NULL");
    std::lock_guard<std::mutex> l(f->m);
    out5[0] = (double)f->n_fed;
    out5[1] = f->n_fed ? 1e6 * f->t_gather / f->n_fed : 0.0;
    out5[2] = f->n_fed ? 1e6 * f->t_copy_call / f->n_fed : 0.0;
    out5[3] = (double)f->n_acquired;
    out5[4] = f->n_acquired ? 1e6 * f->t_acquire_wait / f->n_acquired : 0.0;
    f->t_gather = f->t_copy_call = f->t_acquire_wait = 0;
    f->n_fed = f->n_acquired = 0;
    return MDBN_OK;
}

int mdbn_feeder_cancel(mdbn_feeder* f)
{
    REQUIRE(f != nullptr, "feeder is NULL");
    std::unique_lock<std::mutex> l(f->m);
    f->queue.clear();
    f->cv.wait(l, [&] { return f->inflight.empty(); });
    for (auto& kv : f->ready) f->busy[kv.second.slot] = 0;     // uploaded (or failed) and never read
    f->ready.clear();
    l.unlock();
    f->cv.notify_all();
    return MDBN_OK;
}

int mdbn_feeder_destroy(mdbn_feeder* f)
{
    if (!f) return MDBN_OK;
    {
        std::lock_guard<std::mutex> l(f->m);
        f->stop = true;
    }
    f->cv.notify_all();
    if (f->dispatcher.joinable()) f->dispatcher.join();
    if (f->copier.joinable()) f->copier.join();
    delete f->pool;
    (void)hipSetDevice(f->device);
    if (f->copy_stream) (void)hipStreamSynchronize(f->copy_stream);
    if (f->copy_stream2) (void)hipStreamSynchronize(f->copy_stream2);
    if (f->half_done) (void)hipEventDestroy(f->half_done);
    if (f->half_go) (void)hipEventDestroy(f->half_go);
    if (f->copy_stream2) (void)hipStreamDestroy(f->copy_stream2);
    for (float* st : f->staging) (void)hipHostFree(st);
    for (hipEvent_t ev : f->copied) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : f->consumed) (void)hipEventDestroy(ev);
    if (f->copy_stream) (void)hipStreamDestroy(f->copy_stream);
    delete f;
    return MDBN_OK;
}

}  // extern "C"
