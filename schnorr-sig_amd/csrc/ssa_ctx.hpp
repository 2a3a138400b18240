// Internal: context layout and helpers shared by the translation units of the library
// (ssa_api.hip: per-lane verification path; ssa_msm.hip: MSM-form batch verification).
#pragma once
#include "../../include/schnorr_sig_amd.h"
#include "ssa_kernels.hpp"

#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <thread>
#include <string>
#include <vector>

using namespace ssa;

// a library does not write to stderr on its own: the failing call is reported only when SSA_DEBUG is set
static inline bool ssa_debug_enabled() {
    static const bool on = std::getenv("SSA_DEBUG") != nullptr;
    return on;
}
#define HIP_TRY(expr)                                                                    \
    do {                                                                                 \
        hipError_t err__ = (expr);                                                       \
        if (err__ != hipSuccess) {                                                       \
            if (ssa_debug_enabled())                                                     \
                std::fprintf(stderr, "[schnorr_sig_amd] %s failed: %s (%s:%d)\n", #expr, \
                             hipGetErrorString(err__), __FILE__, __LINE__);              \
            return SSA_ERR_HIP;                                                          \
        }                                                                                \
    } while (0)

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes) {
        if (bytes <= cap) return 0;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        size_t want = bytes + bytes / 8 + 256;
        if (hipMalloc(&p, want) != hipSuccess) return SSA_ERR_HIP;
        cap = want;
        return 0;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

// page-locked HOST memory owned by the library (hipHostMalloc): what the DMA engines read and write in the host-buffer entry
// points.  The caller's memory itself is never registered with the runtime (see pipelined_upload_hash).
struct HostBuf {
    void *p = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes) {
        if (bytes <= cap) return 0;
        if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
        const size_t want = bytes + bytes / 8 + 4096;
        if (hipHostMalloc(&p, want, hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError();
            p = nullptr;
            return SSA_ERR_HIP;
        }
        cap = want;
        return 0;
    }
    void release() {
        if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
    }
};

struct TimedLaunch {
    hipEvent_t start, stop;
};

struct ssa_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    hipStream_t copy_stream = nullptr;        // uploads of the host-buffer entry points, overlapped with the kernels
    hipEvent_t copy_done[8] = {};             // one per upload chunk
    hipStream_t hash_stream[2] = {};          // the chunks' hash launches alternate between two streams, so that the
    hipEvent_t hash_done[8] = {};             //   tail of one launch (a lane hashes for ~4 ms) overlaps the next
    hipEvent_t pipe_start = nullptr;          // everything queued on `stream` before a pipelined upload began
    hipEvent_t order_ev = nullptr;            // ssa_ctx_stream_release / _acquire
    size_t pipeline_min_n = 1 << 17;          // host-buffer batches from this size on are uploaded in chunks
    unsigned pipeline_chunks = 8;             // SSA_PIPELINE_CHUNKS overrides (1 = off)
    DevParams *d_params = nullptr;
    u64 *d_gtab = nullptr;                    // the comb table for G: owned by gtab_share (one per device, generator and
    struct SharedGtab *gtab_share = nullptr;  //   geometry); its first word carries the geometry (ssa_kernels.hpp)
    uint32_t gtab_bits = 0;                   // window width of that table (16 / 20 / 22 / 24)
    uint64_t hbm_budget = 0;                  // bytes the speed-for-memory tables may take (comb for G, per-key combs)
    DevParams h_params;                       // host copy of the blob the context was created from (derived flags set)
    ssa_ctx *twin = nullptr;                  // second set of streams and workspaces: calls of more than one slice
                                              //   alternate their slices between the two (created at the first such call)
    bool is_twin = false, two_streams = true; // SSA_TWO_STREAMS=0 turns the alternation off
    DevBuf ws_h, ws_tab, ws_fail;
    // staging for the host-buffer entry points
    DevBuf st_sigs, st_pks, st_inf, st_msgs, st_off, st_status, st_aux, st_aux2;
    // MSM-form batch verification (ssa_msm.hip)
    DevBuf msm_points, msm_scalars, msm_keys, msm_vals, msm_keys2, msm_vals2, msm_sort_tmp, msm_bounds,
        msm_buckets, msm_chunks, msm_windows, msm_partials, msm_flags, st_coeffs, msm_cnt, msm_cnt2, msm_ids, msm_ids2,
        msm_comb_pts, msm_comb_lins;
    bool timing = false;
    bool default_params = false;   // created from the built-in (unpinned) blob
    // batches up to these sizes take the cooperative (waves-per-signature) kernel: measured crossovers without /
    // with the subgroup check (tools/mode_crossover.py); SSA_COOP_MAX_N overrides both
    size_t coop_max_n = 7680, coop_max_n_torsion = 10496;   // lane kernels: 3.5 / 5.4 ms flat up to 2^15 (round 2, window asm)
    size_t msm_small_max = 3072;  // MSM-form batches up to this size: one cooperative block per signature (measured
                                  // crossover with the bucket method: tools/msm_small_crossover.py; SSA_MSM_SMALL_MAX)
    int fault_after_chunk = -1;   // ssa_debug_fault_after_chunk (tests)
    // Workspace bound: the per-lane kernels run over slices of at most lane_slice lanes (2 KB of table + 32 B of scalar
    // each: 2.1 GB at 2^20), the MSM-form pipeline over slices of at most msm_slice signatures whose records are
    // combined like the shards of a multi-GPU batch.  SSA_LANE_SLICE / SSA_MSM_SLICE override (tests force small ones).
    size_t lane_slice = (size_t)1 << 20, msm_slice = (size_t)1 << 23;
    DevBuf msm_slice_recs;        // one 24-word record per MSM slice
    DevBuf msm_sbuf;              // the coefficients s_i between the two halves of the preparation (32 B per signature)
    bool msm_overlap = true;      // the h-independent half of msm_k_prepare runs under ssa_k_hash (SSA_MSM_OVERLAP=0: off)
    unsigned msm_tree_group = 16; // chunk sums added per cooperating wave and tree level (SSA_MSM_TREE_GROUP: 2..64)
    // signing (ssa_sign.hip): the 4-bit comb table of the constant-time signer (98 KB, built at the first use) and the
    // intermediates of the keyed (130-byte) output
    DevBuf ctab, sg_sigs, sg_pks;
    bool ctab_ready = false;
    unsigned verify_block = 256;  // threads per block of ssa_k_verify (SSA_VERIFY_BLOCK overrides: 64/128/256)
    // the end game of ssa_k_verify (ssa_kernels.hpp "The end game of a launch"): the last generation of lanes runs in
    // tail_pieces pieces per ladder pass, only the last of which stand at the end of the grid
    unsigned verify_waves = 0;    // waves of ssa_k_verify resident at once on this device (occupancy x CUs x 4)
    unsigned tail_pieces = 5;     // SSA_TAIL_PIECES (0 or 1: off); launches of less than one generation have none
    unsigned tail_gens = 1;       // SSA_TAIL_GENS: tail groups, in generations of resident waves
    bool tail_uniform = false;    // SSA_TAIL_UNIFORM=1: equal pieces instead of 1/2, 1/4, 1/8, ...
    bool tail_reversed = false;   // SSA_TAIL_REVERSED=1 (tests): the end game's roles dealt from the end of the grid
    unsigned tail_min_main = 0;   // SSA_TAIL_MIN_MAIN: generations of ordinary workgroups a launch must have beside its tail
    unsigned tail_waves_override = 0;   // SSA_TAIL_WAVES: the tests' small "generation" (the end game on batches of thousands)
    DevBuf tail_done, tail_park;  // per tail group: finished pieces; parked accumulators + status (152 B per lane)
    // page-locked bounce buffers of the host-buffer entry points: one slice of inputs (257 B + message per lane), the
    // caller's coefficients of the MSM form, one slice of statuses
    HostBuf pin_in, pin_coeffs, pin_out;
    std::map<std::string, std::vector<TimedLaunch>> timed;
    std::vector<struct ssa_keyset *> keysets;   // live key sets of this context (orphaned, not leaked, by ssa_ctx_destroy)
};

static inline unsigned grid_for(size_t n, unsigned block) { return (unsigned)((n + block - 1) / block); }

template <class F>
static inline int timed_launch(ssa_ctx *ctx, const char *name, F &&launch) {
    if (!ctx->timing) {
        launch();
        HIP_TRY(hipGetLastError());
        return 0;
    }
    TimedLaunch t;
    HIP_TRY(hipEventCreate(&t.start));
    HIP_TRY(hipEventCreate(&t.stop));
    HIP_TRY(hipEventRecord(t.start, ctx->stream));
    launch();
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(t.stop, ctx->stream));
    ctx->timed[name].push_back(t);
    return 0;
}


// ---- argument checks and host->device staging shared by the entry points ----
static inline int check_msgs(const uint8_t *msgs, const uint64_t *off, size_t stride, size_t len, size_t n) {
    if (n == 0) return 0;
    if (n > SSA_MAX_BATCH) return SSA_ERR_ARG;   // grid sizes and workspace offsets are computed for n <= 2^30
    if (!off && len > 0 && !msgs) return SSA_ERR_ARG;
    if (!off && stride < len) return SSA_ERR_ARG;
    if (len > 0xffffffffull) return SSA_ERR_ARG;
    return 0;
}

static inline size_t msgs_bytes(const uint64_t *off, size_t stride, size_t len, size_t n) {
    if (n == 0) return 0;
    if (off) return (size_t)off[n];
    return (n - 1) * stride + len;
}

struct StagedInputs {
    const u8 *sigs = nullptr, *pks = nullptr, *inf = nullptr, *msgs = nullptr;
    const u64 *off = nullptr;
};

static inline int stage_up(ssa_ctx *ctx, DevBuf &buf, const void *src, size_t bytes, const void **dst) {
    *dst = nullptr;
    if (!src || bytes == 0) {
        if (buf.reserve(16)) return SSA_ERR_HIP;  // non-null dummy for zero-length messages
        *dst = src ? buf.p : nullptr;
        return 0;
    }
    if (buf.reserve(bytes)) return SSA_ERR_HIP;
    HIP_TRY(hipMemcpyAsync(buf.p, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    *dst = buf.p;
    return 0;
}

static inline int stage_msgs(ssa_ctx *ctx, const uint8_t *msgs, const uint64_t *off, size_t stride, size_t len,
                      size_t n, StagedInputs &s) {
    const void *p;
    if (off) {
        for (size_t i = 0; i < n; i++)
            if (off[i + 1] < off[i] || off[i + 1] - off[i] > 0xffffffffull) return SSA_ERR_ARG;
        if (int rc = stage_up(ctx, ctx->st_off, off, (n + 1) * sizeof(uint64_t), &p)) return rc;
        s.off = (const u64 *)p;
    }
    const size_t mb = msgs_bytes(off, stride, len, n);
    if (mb && !msgs) return SSA_ERR_ARG;
    if (ctx->st_msgs.reserve(mb + 16)) return SSA_ERR_HIP;
    if (mb) HIP_TRY(hipMemcpyAsync(ctx->st_msgs.p, msgs, mb, hipMemcpyHostToDevice, ctx->stream));
    s.msgs = (const u8 *)ctx->st_msgs.p;
    return 0;
}


// defined in ssa_api.hip
int ssa_internal_hash_chunk(ssa_ctx *ctx, hipStream_t hs, const uint8_t *d_sigs, const uint8_t *d_pks, const uint8_t *d_msgs,
                            const uint64_t *d_off, size_t msg_stride, size_t msg_len, size_t cnt, uint64_t *d_h);

// Host-buffer uploads go through page-locked memory of the LIBRARY's (ctx->pin_in): the caller's bytes are copied into it
// by host threads, chunk by chunk, and the DMA engines read it asynchronously, under the kernels of the chunk before.
// Rounds 3-5 registered the CALLER's memory with the runtime for the duration of a call instead (hipHostRegister: no host
// copy, 29 M verifications/s).  That is not safe: for buffers that live in the process heap -- where glibc puts even
// megabyte arrays once its mmap threshold has grown, and where a range shares its first and last page with its neighbours --
// a later ordinary copy out of such memory faulted on the GPU side (`Memory access fault` on a page-aligned heap address;
// tools/soak_large.py found it within ten iterations, never with the in-place registration off, never with the arrays in
// their own mappings).  Whatever the runtime keeps of a registration, a library has no business changing the mapping state
// of memory it does not own.
struct PipelinedInputs {
    StagedInputs s;     // device copies (context staging buffers)
    ssa_ctx *armed = nullptr;   // set with the first enqueue; cleared by done() once the call has synchronised
    void done() { armed = nullptr; }
    ~PipelinedInputs() {
        // an error return after the first enqueue: copies out of the bounce buffers and hash launches that write
        // ctx->ws_h may still be in flight -- wait for them before the next call reuses the buffers
        if (!armed) return;
        (void)hipStreamSynchronize(armed->copy_stream);
        for (auto &hs : armed->hash_stream) (void)hipStreamSynchronize(hs);
        (void)hipStreamSynchronize(armed->stream);
    }
};

// caller memory -> page-locked memory on up to SSA_COPY_THREADS (default 8) host threads (one thread moves ~10 GB/s; a
// 2^20-signature slice is 270 MB and its upload must not take as long as its kernels)
static inline size_t host_copy_threads() {
    static const size_t t = [] {
        const char *e = std::getenv("SSA_COPY_THREADS");
        const long v = e ? std::atol(e) : 8;
        return (size_t)(v < 1 ? 1 : v > 32 ? 32 : v);
    }();
    return t;
}
static inline void host_copy(void *dst, const void *src, size_t bytes) {
    constexpr size_t PIECE = 4u << 20;
    if (bytes <= 2 * PIECE || host_copy_threads() == 1) {
        std::memcpy(dst, src, bytes);
        return;
    }
    const size_t parts = bytes / PIECE < host_copy_threads() ? bytes / PIECE : host_copy_threads();
    const size_t per = (bytes / parts + 63) & ~(size_t)63;
    std::vector<std::thread> th;
    for (size_t t = 1; t < parts; t++) {
        const size_t lo = t * per, hi = t + 1 == parts ? bytes : (t + 1) * per;
        th.emplace_back([=] { std::memcpy((char *)dst + lo, (const char *)src + lo, hi - lo); });
    }
    std::memcpy(dst, src, per < bytes ? per : bytes);
    for (auto &x : th) x.join();
}

// debug hook of the error-path tests (ssa_debug_fault_after_chunk): the next pipelined upload fails (SSA_ERR_HIP) after
// chunk k has been enqueued.  One shot, armed through the ABI on this context only: no environment is read per call.
static inline int pipeline_fault_chunk(ssa_ctx *ctx) {
    const int k = ctx->fault_after_chunk;
    ctx->fault_after_chunk = -1;
    return k;
}

// Uploads of a large host-buffer batch in chunks on the copy stream; the challenge hashes of chunk c start as soon as
// chunk c has arrived (they are 27 % of the per-signature work, 62 % of the MSM form), alternating between two
// streams -- a lane hashes for ~4 ms and a launch's tail would otherwise idle most of the chip once per chunk.
// Ordering: the side streams first wait for everything already queued on ctx->stream (an earlier asynchronous
// *_device call may still read ws_h or the staging buffers this call overwrites), and on return ctx->stream waits for
// all of it: whatever the caller enqueues next sees the inputs and ctx->ws_h.
// *used == false: the ranges could not be pinned (e.g. a read-only mapping) and nothing was enqueued.
static inline int pipelined_upload_hash(ssa_ctx *ctx, const uint8_t *sigs, const uint8_t *pks, const uint8_t *pk_inf,
                                        const uint8_t *msgs, const uint64_t *msg_off, size_t msg_stride, size_t msg_len,
                                        size_t n, PipelinedInputs &pin, bool *used) {
    *used = false;
    // arguments first: nothing is pinned or enqueued for a call that is going to be refused
    const size_t mb = msgs_bytes(msg_off, msg_stride, msg_len, n);
    if (mb && !msgs) return SSA_ERR_ARG;
    if (msg_off)
        for (size_t i = 0; i < n; i++)
            if (msg_off[i + 1] < msg_off[i] || msg_off[i + 1] - msg_off[i] > 0xffffffffull) return SSA_ERR_ARG;
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t o_pks = al(n * 81), o_msgs = o_pks + al(n * 96), o_inf = o_msgs + al(mb),
                 o_off = o_inf + al(pk_inf ? n : 0), total = o_off + al(msg_off ? (n + 1) * sizeof(uint64_t) : 0);
    if (ctx->pin_in.reserve(total)) return 0;      // no page-locked memory to be had: the staged path
    u8 *h_sigs = (u8 *)ctx->pin_in.p, *h_pks = h_sigs + o_pks, *h_msgs = h_sigs + o_msgs, *h_inf = h_sigs + o_inf,
       *h_off = h_sigs + o_off;
    *used = true;
    if (ctx->st_sigs.reserve(n * 81) || ctx->st_pks.reserve(n * 96) || ctx->st_msgs.reserve(mb + 16) ||
        ctx->ws_h.reserve(n * 4 * sizeof(u64)) || (msg_off && ctx->st_off.reserve((n + 1) * sizeof(uint64_t))) ||
        (pk_inf && ctx->st_inf.reserve(n)))
        return SSA_ERR_HIP;
    HIP_TRY(hipEventRecord(ctx->pipe_start, ctx->stream));
    HIP_TRY(hipStreamWaitEvent(ctx->copy_stream, ctx->pipe_start, 0));
    for (auto &hs : ctx->hash_stream) HIP_TRY(hipStreamWaitEvent(hs, ctx->pipe_start, 0));
    pin.armed = ctx;
    const int fault_chunk = pipeline_fault_chunk(ctx);
    const u64 *d_off = nullptr;
    if (msg_off) {
        host_copy(h_off, msg_off, (n + 1) * sizeof(uint64_t));
        HIP_TRY(hipMemcpyAsync(ctx->st_off.p, h_off, (n + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->copy_stream));
        d_off = (const u64 *)ctx->st_off.p;
    }
    if (pk_inf) {
        host_copy(h_inf, pk_inf, n);
        HIP_TRY(hipMemcpyAsync(ctx->st_inf.p, h_inf, n, hipMemcpyHostToDevice, ctx->copy_stream));
        pin.s.inf = (const u8 *)ctx->st_inf.p;
    }
    const unsigned chunks = ctx->pipeline_chunks;
    u8 *d_sigs = (u8 *)ctx->st_sigs.p, *d_pks = (u8 *)ctx->st_pks.p, *d_msgs = (u8 *)ctx->st_msgs.p;
    for (unsigned c = 0; c < chunks; c++) {
        const size_t lo = n * c / chunks, hi = n * (c + 1) / chunks, cnt = hi - lo;
        if (cnt == 0) continue;
        // (the host copies of chunk c run while the DMA engines and the hash kernels work on chunk c - 1)
        host_copy(h_sigs + 81 * lo, sigs + 81 * lo, cnt * 81);
        HIP_TRY(hipMemcpyAsync(d_sigs + 81 * lo, h_sigs + 81 * lo, cnt * 81, hipMemcpyHostToDevice, ctx->copy_stream));
        host_copy(h_pks + 96 * lo, pks + 96 * lo, cnt * 96);
        HIP_TRY(hipMemcpyAsync(d_pks + 96 * lo, h_pks + 96 * lo, cnt * 96, hipMemcpyHostToDevice, ctx->copy_stream));
        const size_t m_lo = msg_off ? (size_t)msg_off[lo] : lo * msg_stride;
        const size_t m_hi = msg_off ? (size_t)msg_off[hi] : (hi == n ? mb : hi * msg_stride);
        if (m_hi > m_lo) {
            host_copy(h_msgs + m_lo, msgs + m_lo, m_hi - m_lo);
            HIP_TRY(hipMemcpyAsync(d_msgs + m_lo, h_msgs + m_lo, m_hi - m_lo, hipMemcpyHostToDevice, ctx->copy_stream));
        }
        HIP_TRY(hipEventRecord(ctx->copy_done[c], ctx->copy_stream));
        hipStream_t hs = ctx->hash_stream[c & 1u];
        HIP_TRY(hipStreamWaitEvent(hs, ctx->copy_done[c], 0));
        if (int rc = ssa_internal_hash_chunk(ctx, hs, d_sigs + 81 * lo, d_pks + 96 * lo,
                                             msg_off ? d_msgs : d_msgs + lo * msg_stride, msg_off ? d_off + lo : nullptr,
                                             msg_stride, msg_len, cnt, (uint64_t *)ctx->ws_h.p + 4 * lo))
            return rc;
        HIP_TRY(hipEventRecord(ctx->hash_done[c], hs));
        HIP_TRY(hipStreamWaitEvent(ctx->stream, ctx->hash_done[c], 0));
        if ((int)c == fault_chunk) return SSA_ERR_HIP;   // injected (tests)
    }
    pin.s.sigs = d_sigs;
    pin.s.pks = d_pks;
    pin.s.msgs = d_msgs;
    pin.s.off = d_off;
    return 0;
}

// defined in ssa_api.hip: hash_message + Scalar::from_bits_vartime for n signatures into ctx->ws_h
int ssa_internal_hash_scalars(ssa_ctx *ctx, const uint8_t *d_sigs, const uint8_t *d_pks, const uint8_t *d_msgs,
                              const uint64_t *d_msg_off, size_t msg_stride, size_t msg_len, size_t n);

// defined in ssa_api.hip: the context's second set of streams and workspaces (nullptr: none -- a twin itself, turned
// off, or no memory for it)
ssa_ctx *ssa_internal_twin(ssa_ctx *ctx);

// Host-buffer batches in bounded device memory (round 5): more than one slice of lanes runs slice after slice through
// staging buffers sized for ONE slice, pinning only the slice in flight, into the caller's one status array and one
// counter -- fn(c, lo, cnt) is the one-slice form of the entry point on context c.  With a twin (ssa_internal_twin) two host
// threads take alternate slices, one on the context and one on its twin: the upload of a slice runs under the kernels
// of the other, and the kernels' tails fill each other.
template <class F>
static int run_host_slices(ssa_ctx *ctx, size_t n, size_t slice, F &&fn) {
    const size_t k = (n + slice - 1) / slice;
    ssa_ctx *tw = k > 1 ? ssa_internal_twin(ctx) : nullptr;
    if (!tw) {
        for (size_t j = 0; j < k; j++) {
            const size_t lo = j * slice, cnt = n - lo < slice ? n - lo : slice;
            if (int rc = fn(ctx, lo, cnt)) return rc;
        }
        return 0;
    }
    int rcs[2] = {0, 0};
    auto worker = [&](size_t w) {
        if (hipSetDevice(ctx->device) != hipSuccess) {
            rcs[w] = SSA_ERR_HIP;
            return;
        }
        ssa_ctx *c = w ? tw : ctx;
        for (size_t j = w; j < k && rcs[w] == 0; j += 2) {
            const size_t lo = j * slice, cnt = n - lo < slice ? n - lo : slice;
            rcs[w] = fn(c, lo, cnt);
        }
    };
    std::thread second(worker, (size_t)1);
    worker(0);
    second.join();
    return rcs[0] ? rcs[0] : rcs[1];
}

// the messages of lanes [lo, lo + cnt) of a host batch as a batch of their own (an offset table is rebased)
struct HostMsgSlice {
    std::vector<uint64_t> off;
    const uint8_t *msgs = nullptr;
    const uint64_t *offp = nullptr;
    HostMsgSlice(const uint8_t *all, const uint64_t *msg_off, size_t msg_stride, size_t lo, size_t cnt) {
        if (msg_off) {
            off.resize(cnt + 1);
            for (size_t k = 0; k <= cnt; k++) off[k] = msg_off[lo + k] - msg_off[lo];
            msgs = all ? all + msg_off[lo] : nullptr;
            offp = off.data();
        } else {
            msgs = all ? all + lo * msg_stride : nullptr;
        }
    }
};

