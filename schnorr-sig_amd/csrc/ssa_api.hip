// C ABI (include/schnorr_sig_amd.h) over the HIP kernels in ssa_kernels.hpp.
// No CPU compute path exists here: every entry point launches kernels on the context's
// device or fails with an SSA_ERR_* code.
#define SSA_KERNELS_DEFINE 1
#include "ssa_ctx.hpp"

#include <mutex>
#include <thread>

static const unsigned char k_default_params[SSA_PARAMS_LENGTH] = {
#include "../params/params_default.inc"
};

extern "C" const char *ssa_strerror(int rc) {
    switch (rc) {
        case SSA_OK: return "ok";
        case SSA_INVALID_PUBLIC_KEY: return "The public key is not an element of the prime subgroup.";
        case SSA_INVALID_SIGNATURE: return "The signature is invalid or was incorrectly computed.";
        case SSA_MALFORMED: return "malformed input (the reference would panic)";
        case SSA_ERR_ARG: return "invalid argument";
        case SSA_ERR_HIP: return "HIP runtime error";
        case SSA_ERR_PARAMS: return "invalid parameter blob";
        case SSA_ERR_NO_DEVICE: return "no HIP device";
        default: return "unknown";
    }
}

extern "C" const void *ssa_default_params(void) { return k_default_params; }

extern "C" int ssa_abi_version(void) { return SSA_ABI_VERSION; }

// keyed context (defined here because ssa_ctx_destroy orphans the key sets that outlive their context)
struct ssa_keyset {
    ssa_ctx *ctx = nullptr;   // nullptr: the context is gone, the device memory went with it
    size_t m = 0;
    bool comb = false;      // per-key comb tables (16 x 65536 rows = 100 MB per key) instead of the ladder's 16 multiples
    DevBuf tab, status, pks, ktab;
    void release_all() {
        tab.release();
        status.release();
        pks.release();
        ktab.release();
    }
};

// The comb table depends on the device, the generator and its geometry only, and it is up to 17.7 GB: contexts of one
// process share it (reference-counted; ssa_multi_create with several contexts per device, the tests' many engines, a
// binding that makes a context per thread).  Built at the first acquisition on the acquiring context's stream,
// synchronously, under the registry's lock; read-only afterwards.
struct SharedGtab {
    int device = 0;
    u32 bits = 0;
    u64 gen[12] = {};
    u64 *d_gtab = nullptr;
    int refs = 0;
};
static std::mutex g_gtab_mu;
static std::vector<SharedGtab *> g_gtabs;

static inline size_t gtab_bytes(u32 bits) { return gtab_entries(bits) * 12 * sizeof(u64); }

// the table of `bits`-bit windows for this generator on this device: an existing one, or a new one (nullptr: no memory)
static SharedGtab *gtab_acquire(ssa_ctx *ctx, const DevParams &hp, u32 bits) {
    std::lock_guard<std::mutex> lock(g_gtab_mu);
    u64 gen[12];
    for (int i = 0; i < 6; i++) {
        gen[i] = hp.gen_x[i];
        gen[6 + i] = hp.gen_y[i];
    }
    for (SharedGtab *g : g_gtabs)
        if (g->device == ctx->device && g->bits == bits && std::memcmp(g->gen, gen, sizeof gen) == 0) {
            g->refs++;
            return g;
        }
    SharedGtab *g = new SharedGtab();
    g->device = ctx->device;
    g->bits = bits;
    std::memcpy(g->gen, gen, sizeof gen);
    // base entries by double-and-add (90 112 of them for 24-bit windows), then one affine addition per entry
    // (ssa_kernels.hpp)
    void *gbase = nullptr;
    bool ok = hipMalloc((void **)&g->d_gtab, gtab_bytes(bits)) == hipSuccess &&
              hipMalloc(&gbase, gbase_entries(bits) * 12 * sizeof(u64)) == hipSuccess;
    if (ok) {
        hipLaunchKernelGGL(ssa_k_gbase, dim3(grid_for(gbase_entries(bits), 256)), dim3(256), 0, ctx->stream, ctx->d_params,
                           (u64 *)gbase, bits);
        hipLaunchKernelGGL(ssa_k_gtable, dim3(grid_for(gtab_entries(bits) / 8, 256)), dim3(256), 0, ctx->stream,
                           (const u64 *)gbase, g->d_gtab, bits);
        ok = hipGetLastError() == hipSuccess && hipStreamSynchronize(ctx->stream) == hipSuccess;
    } else {
        (void)hipGetLastError();      // an allocation that did not fit is not a sticky error: the caller tries a smaller table
    }
    if (gbase) (void)hipFree(gbase);
    if (!ok) {
        if (g->d_gtab) (void)hipFree(g->d_gtab);
        delete g;
        return nullptr;
    }
    g->refs = 1;
    g_gtabs.push_back(g);
    return g;
}

static void gtab_release(SharedGtab *g) {
    if (!g) return;
    std::lock_guard<std::mutex> lock(g_gtab_mu);
    if (--g->refs > 0) return;
    for (size_t i = 0; i < g_gtabs.size(); i++)
        if (g_gtabs[i] == g) {
            g_gtabs.erase(g_gtabs.begin() + (long)i);
            break;
        }
    (void)hipSetDevice(g->device);
    (void)hipFree(g->d_gtab);
    delete g;
}

static int validate_params(const DevParams &p) {
    if (std::memcmp(p.magic, "SSAPARM1", 8) != 0) return SSA_ERR_PARAMS;
    if (p.n_rounds == 0 || p.n_rounds > 8) return SSA_ERR_PARAMS;
    if (p.rate_off != 0 && p.rate_off != 4) return SSA_ERR_PARAMS;
    if (p.cap_len_idx < -1 || p.cap_len_idx > 11) return SSA_ERR_PARAMS;
    if (p.pad_mode > 1 || p.digest_off > 8) return SSA_ERR_PARAMS;
    for (int i = 0; i < 144; i++)
        if (p.mds[i] >= FP_P) return SSA_ERR_PARAMS;
    for (int i = 0; i < 96; i++)
        if (p.ark1[i] >= FP_P || p.ark2[i] >= FP_P) return SSA_ERR_PARAMS;
    for (int i = 0; i < 6; i++)
        if (p.gen_x[i] >= FP_P || p.gen_y[i] >= FP_P) return SSA_ERR_PARAMS;
    return 0;
}

extern "C" int ssa_ctx_create(ssa_ctx **out, int device, const void *params, size_t params_len) {
    return ssa_ctx_create_ex(out, device, params, params_len, 0u, 0ull);
}

// gtab_bits: window width of the comb for G (16, 20, 22 or 24; 0 = the widest whose table fits the budget; the
// environment's SSA_GTAB_BITS overrides 0).  hbm_budget_bytes: what the library may spend on tables that are a pure
// speed-for-memory trade (the comb for G, per-key combs of a key set); 0 = a tenth of the device memory that is free
// when the context is created (SSA_HBM_BUDGET_MB overrides 0).  An allocation that fails falls back to the next
// smaller table instead of failing the context: the 16-bit comb (100 MB) is the floor.
extern "C" int ssa_ctx_create_ex(ssa_ctx **out, int device, const void *params, size_t params_len, uint32_t gtab_bits,
                                 uint64_t hbm_budget_bytes) {
    if (!out) return SSA_ERR_ARG;
    *out = nullptr;
    if (gtab_bits != 0 && gtab_bits != 16 && gtab_bits != 20 && gtab_bits != 22 && gtab_bits != 24) return SSA_ERR_ARG;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count == 0) return SSA_ERR_NO_DEVICE;
    if (device < 0 || device >= count) return SSA_ERR_ARG;
    DevParams hp;
    if (params) {
        if (params_len != sizeof(DevParams)) return SSA_ERR_PARAMS;
        std::memcpy(&hp, params, sizeof hp);
    } else {
        std::memcpy(&hp, k_default_params, sizeof hp);
    }
    // the built-in blob is the builder's own instance (Rescue constants and generator are NOT upstream's: DESIGN.md
    // "parity unpinned"); a context created from it says so (ssa_ctx_uses_default_params)
    const bool is_default = std::memcmp(&hp, k_default_params, sizeof hp) == 0;
    if (int rc = validate_params(hp)) return rc;
    hp.flags = 0;  // derived flags are the library's, not the caller's
    bool small_mds = true;
    for (int i = 0; i < 144; i++) small_mds = small_mds && hp.mds[i] <= 0xffffffffull;
    if (small_mds) hp.flags |= PRM_FLAG_SMALL_MDS;
    bool tiny_mds = true;      // entries below 2^16 (the usual circulant of single-digit integers): carry-free MDS rows
    for (int i = 0; i < 144; i++) tiny_mds = tiny_mds && hp.mds[i] < 0x10000ull;
    if (tiny_mds) hp.flags |= PRM_FLAG_TINY_MDS;
    HIP_TRY(hipSetDevice(device));
    ssa_ctx *ctx = new ssa_ctx();
    ctx->device = device;
    ctx->default_params = is_default;
    if (const char *cm = std::getenv("SSA_COOP_MAX_N"))
        ctx->coop_max_n = ctx->coop_max_n_torsion = (size_t)std::strtoull(cm, nullptr, 10);
    if (const char *sm = std::getenv("SSA_MSM_SMALL_MAX")) ctx->msm_small_max = (size_t)std::strtoull(sm, nullptr, 10);
    if (const char *vb = std::getenv("SSA_VERIFY_BLOCK")) {
        const int v = std::atoi(vb);
        if (v == 64 || v == 128 || v == 256) ctx->verify_block = (unsigned)v;
    }
    if (const char *ls = std::getenv("SSA_LANE_SLICE")) {      // lanes per slice of the per-lane kernels (workspace bound)
        const size_t v = (size_t)std::strtoull(ls, nullptr, 10);
        if (v >= 256) ctx->lane_slice = v;
    }
    if (const char *ms = std::getenv("SSA_MSM_SLICE")) {       // signatures per slice of the MSM-form pipeline
        const size_t v = (size_t)std::strtoull(ms, nullptr, 10);
        if (v >= 256 && v <= ((size_t)1 << 23)) ctx->msm_slice = v;
    }
    if (const char *tg = std::getenv("SSA_MSM_TREE_GROUP")) {
        const int v = std::atoi(tg);
        if (v >= 2 && v <= 64) ctx->msm_tree_group = (unsigned)v;
    }
    if (const char *tpc = std::getenv("SSA_TAIL_PIECES")) {    // pieces of a tail group's work (0 / 1: no end game)
        const int v = std::atoi(tpc);
        if (v >= 0 && v <= VP_MAX) ctx->tail_pieces = (unsigned)v;
    }
    if (const char *tg = std::getenv("SSA_TAIL_GENS")) {       // tail groups, in generations of resident waves
        const int v = std::atoi(tg);
        if (v >= 1 && v <= 8) ctx->tail_gens = (unsigned)v;
    }
    if (const char *tu = std::getenv("SSA_TAIL_UNIFORM")) ctx->tail_uniform = std::atoi(tu) != 0;
    if (const char *tm = std::getenv("SSA_TAIL_MIN_MAIN")) ctx->tail_min_main = (unsigned)std::atoi(tm);
    if (const char *tr = std::getenv("SSA_TAIL_REVERSED")) ctx->tail_reversed = std::atoi(tr) != 0;
    if (const char *tw = std::getenv("SSA_TAIL_WAVES")) ctx->tail_waves_override = (unsigned)std::atoi(tw);   // tests: a small "generation"
    if (const char *ts = std::getenv("SSA_TWO_STREAMS")) ctx->two_streams = std::atoi(ts) != 0;
    if (const char *mo = std::getenv("SSA_MSM_OVERLAP")) ctx->msm_overlap = std::atoi(mo) != 0;
    if (const char *pc = std::getenv("SSA_PIPELINE_CHUNKS")) {
        const int v = std::atoi(pc);
        if (v >= 1 && v <= 8) ctx->pipeline_chunks = (unsigned)v;
    }
    if (hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking) != hipSuccess) {
        ssa_ctx_destroy(ctx);
        return SSA_ERR_HIP;
    }
    for (auto &st : ctx->hash_stream)
        if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) {
            ssa_ctx_destroy(ctx);
            return SSA_ERR_HIP;
        }
    if (hipEventCreateWithFlags(&ctx->pipe_start, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->order_ev, hipEventDisableTiming) != hipSuccess) {
        ssa_ctx_destroy(ctx);
        return SSA_ERR_HIP;
    }
    for (int i = 0; i < 8; i++)
        if (hipEventCreateWithFlags(&ctx->copy_done[i], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&ctx->hash_done[i], hipEventDisableTiming) != hipSuccess) {
            ssa_ctx_destroy(ctx);
            return SSA_ERR_HIP;
        }
    ctx->stream = ctx->own_stream;
    {   // the waves of ssa_k_verify that are resident at once: the size of its end game
        hipDeviceProp_t prop;
        int occ = 0;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess &&
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, ssa_k_verify, 256, 0) == hipSuccess && occ > 0)
            ctx->verify_waves = (unsigned)prop.multiProcessorCount * (unsigned)occ * 4u;
        (void)hipGetLastError();
        if (ctx->tail_waves_override) ctx->verify_waves = ctx->tail_waves_override;
    }
    if (hipMalloc((void **)&ctx->d_params, sizeof(DevParams)) != hipSuccess) {
        ssa_ctx_destroy(ctx);
        return SSA_ERR_HIP;
    }
    if (hipMemcpy(ctx->d_params, &hp, sizeof hp, hipMemcpyHostToDevice) != hipSuccess) {
        ssa_ctx_destroy(ctx);
        return SSA_ERR_HIP;
    }
    ctx->h_params = hp;
    // HBM budget of the speed-for-memory tables, and the comb geometry it allows
    if (hbm_budget_bytes == 0) {
        if (const char *mb = std::getenv("SSA_HBM_BUDGET_MB")) hbm_budget_bytes = std::strtoull(mb, nullptr, 10) << 20;
    }
    if (hbm_budget_bytes == 0) {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = 0;
        hbm_budget_bytes = free_b / 10;
    }
    ctx->hbm_budget = hbm_budget_bytes;
    if (gtab_bits == 0) {
        if (const char *gb = std::getenv("SSA_GTAB_BITS")) {
            const int v = std::atoi(gb);
            if (v == 16 || v == 20 || v == 22 || v == 24) gtab_bits = (uint32_t)v;
        }
    }
    // the comb table of this generator on this device: shared by every context that asks for the same geometry.
    // Forced width: that one first; automatic: the widest within the budget.  Either way a failed allocation moves on
    // to the next smaller table.
    static const u32 k_widths[4] = {24, 22, 20, 16};
    for (u32 wbits : k_widths) {
        if (gtab_bits ? wbits > gtab_bits : (wbits > 16 && gtab_bytes(wbits) > hbm_budget_bytes)) continue;
        ctx->gtab_share = gtab_acquire(ctx, hp, wbits);
        if (ctx->gtab_share) break;
    }
    if (!ctx->gtab_share || ctx->ws_fail.reserve(64)) {
        ssa_ctx_destroy(ctx);
        return SSA_ERR_HIP;
    }
    ctx->d_gtab = ctx->gtab_share->d_gtab;
    ctx->gtab_bits = ctx->gtab_share->bits;
    // the generator must be a point of the prime-order subgroup: on the curve, [q]G == O (through the comb table
    // just built), G != O -- otherwise every verification would run on some other curve or a small subgroup
    unsigned gen_ok = 0;
    hipLaunchKernelGGL(ssa_k_check_generator, dim3(1), dim3(64), 0, ctx->stream, ctx->d_params,
                       (const u64 *)ctx->d_gtab, (unsigned *)ctx->ws_fail.p);
    if (hipGetLastError() != hipSuccess ||
        hipMemcpyAsync(&gen_ok, ctx->ws_fail.p, sizeof gen_ok, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
        hipStreamSynchronize(ctx->stream) != hipSuccess) {
        ssa_ctx_destroy(ctx);
        return SSA_ERR_HIP;
    }
    if (gen_ok != 1u) {
        ssa_ctx_destroy(ctx);
        return SSA_ERR_PARAMS;
    }
    *out = ctx;
    return 0;
}

// The second set of streams and workspaces of a context (calls of more than one slice alternate between the two, so
// that the tail of one slice's kernels -- the last wave of every SIMD runs alone, the XCDs finish 1.5-4 % apart: 0.9 ms
// of a 26.6 ms ssa_k_verify, 0.35 of an 8 ms ssa_k_hash -- is filled by the next slice's): a context of its own on the
// same device, blob and comb table (the registry hands the table out again: no second copy), owned by `ctx`.
ssa_ctx *ssa_internal_twin(ssa_ctx *ctx) {
    if (ctx->is_twin || !ctx->two_streams) return nullptr;
    if (!ctx->twin) {
        DevParams hp = ctx->h_params;
        hp.flags = 0;
        ssa_ctx *t = nullptr;
        if (ssa_ctx_create_ex(&t, ctx->device, &hp, sizeof hp, ctx->gtab_bits, ctx->hbm_budget) != 0) {
            ctx->two_streams = false;          // no memory for a second workspace: one stream, as before
            return nullptr;
        }
        t->is_twin = true;
        t->default_params = ctx->default_params;
        t->lane_slice = ctx->lane_slice;
        t->coop_max_n = ctx->coop_max_n;
        t->coop_max_n_torsion = ctx->coop_max_n_torsion;
        t->pipeline_chunks = ctx->pipeline_chunks;
        t->pipeline_min_n = ctx->pipeline_min_n;
        t->verify_block = ctx->verify_block;
        t->tail_pieces = ctx->tail_pieces;
        t->tail_gens = ctx->tail_gens;
        t->tail_uniform = ctx->tail_uniform;
        t->tail_min_main = ctx->tail_min_main;
        t->tail_reversed = ctx->tail_reversed;
        t->verify_waves = ctx->verify_waves;
        ctx->twin = t;
    }
    ctx->twin->timing = ctx->timing;
    return ctx->twin;
}

extern "C" void ssa_ctx_destroy(ssa_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    if (ctx->twin) {
        ssa_ctx_destroy(ctx->twin);
        ctx->twin = nullptr;
    }
    // a key set that outlives its context (garbage-collection order of a binding) must not touch the dead stream:
    // its tables are freed here, the handle stays valid for ssa_keyset_destroy and is refused everywhere else
    for (ssa_keyset *ks : ctx->keysets) {
        ks->release_all();
        ks->ctx = nullptr;
    }
    ctx->keysets.clear();
    for (auto &kv : ctx->timed)
        for (auto &t : kv.second) {
            (void)hipEventDestroy(t.start);
            (void)hipEventDestroy(t.stop);
        }
    for (DevBuf *b : {&ctx->ws_h, &ctx->ws_tab, &ctx->ws_fail, &ctx->st_sigs, &ctx->st_pks, &ctx->st_inf,
                      &ctx->st_msgs, &ctx->st_off, &ctx->st_status, &ctx->st_aux, &ctx->st_aux2, &ctx->msm_points,
                      &ctx->msm_scalars, &ctx->msm_keys, &ctx->msm_vals, &ctx->msm_keys2, &ctx->msm_vals2,
                      &ctx->msm_sort_tmp, &ctx->msm_bounds, &ctx->msm_buckets, &ctx->msm_chunks, &ctx->msm_windows,
                      &ctx->msm_partials, &ctx->msm_flags, &ctx->st_coeffs, &ctx->msm_cnt, &ctx->msm_cnt2,
                      &ctx->msm_ids, &ctx->msm_ids2, &ctx->msm_comb_pts, &ctx->msm_comb_lins, &ctx->msm_slice_recs, &ctx->msm_sbuf, &ctx->tail_done, &ctx->tail_park, &ctx->ctab, &ctx->sg_sigs, &ctx->sg_pks})
        b->release();
    for (HostBuf *b : {&ctx->pin_in, &ctx->pin_coeffs, &ctx->pin_out}) b->release();
    if (ctx->d_params) (void)hipFree(ctx->d_params);
    gtab_release(ctx->gtab_share);
    ctx->gtab_share = nullptr;
    ctx->d_gtab = nullptr;
    for (auto &ev : ctx->copy_done)
        if (ev) (void)hipEventDestroy(ev);
    for (auto &ev : ctx->hash_done)
        if (ev) (void)hipEventDestroy(ev);
    if (ctx->pipe_start) (void)hipEventDestroy(ctx->pipe_start);
    if (ctx->order_ev) (void)hipEventDestroy(ctx->order_ev);
    for (auto &st : ctx->hash_stream)
        if (st) (void)hipStreamDestroy(st);
    if (ctx->copy_stream) (void)hipStreamDestroy(ctx->copy_stream);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
}

// What the context holds on the device: out[0] window bits and out[1] windows of the comb for G, out[2] its bytes
// (shared by the contexts of a process), out[3] bytes of this context's workspaces and staging buffers as reserved so
// far (its second stream's included), out[4] lanes per slice of the per-lane kernels, out[5] signatures per slice of
// the MSM form, out[6] the HBM budget of the speed-for-memory tables, out[7] 1 when slices alternate between two streams.
extern "C" int ssa_ctx_info(const ssa_ctx *ctx, uint64_t out[8]) {
    if (!ctx || !out) return SSA_ERR_ARG;
    auto reserved = [](const ssa_ctx *c) {
        uint64_t sum = 0;
        for (const DevBuf *b : {&c->ws_h, &c->ws_tab, &c->ws_fail, &c->st_sigs, &c->st_pks, &c->st_inf, &c->st_msgs,
                                &c->st_off, &c->st_status, &c->st_aux, &c->st_aux2, &c->msm_points, &c->msm_scalars,
                                &c->msm_keys, &c->msm_vals, &c->msm_keys2, &c->msm_vals2, &c->msm_sort_tmp, &c->msm_bounds,
                                &c->msm_buckets, &c->msm_chunks, &c->msm_windows, &c->msm_partials, &c->msm_flags,
                                &c->st_coeffs, &c->msm_cnt, &c->msm_cnt2, &c->msm_ids, &c->msm_ids2, &c->msm_comb_pts,
                                &c->msm_comb_lins, &c->msm_slice_recs, &c->msm_sbuf, &c->tail_done, &c->tail_park, &c->ctab, &c->sg_sigs, &c->sg_pks})
            sum += b->cap;
        return sum;
    };
    out[0] = ctx->gtab_bits;
    out[1] = gtab_windows(ctx->gtab_bits);
    out[2] = gtab_bytes(ctx->gtab_bits);
    out[3] = reserved(ctx) + (ctx->twin ? reserved(ctx->twin) : 0);
    out[4] = ctx->lane_slice;
    out[5] = ctx->msm_slice;
    out[6] = ctx->hbm_budget;
    out[7] = ctx->two_streams && !ctx->is_twin ? 1 : 0;
    return 0;
}

extern "C" int ssa_ctx_uses_default_params(const ssa_ctx *ctx) { return ctx ? (ctx->default_params ? 1 : 0) : SSA_ERR_ARG; }

extern "C" int ssa_ctx_set_stream(ssa_ctx *ctx, void *hip_stream) {
    if (!ctx) return SSA_ERR_ARG;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
    return 0;
}

// Ordering between the context's stream and a stream of the caller, without a host synchronisation (one event each way).
extern "C" int ssa_ctx_stream_release(ssa_ctx *ctx, void *consumer_stream) {
    if (!ctx) return SSA_ERR_ARG;
    HIP_TRY(hipSetDevice(ctx->device));
    if ((hipStream_t)consumer_stream == ctx->stream) return 0;      // same stream: already ordered
    HIP_TRY(hipEventRecord(ctx->order_ev, ctx->stream));
    HIP_TRY(hipStreamWaitEvent((hipStream_t)consumer_stream, ctx->order_ev, 0));
    return 0;
}

extern "C" int ssa_ctx_stream_acquire(ssa_ctx *ctx, void *producer_stream) {
    if (!ctx) return SSA_ERR_ARG;
    HIP_TRY(hipSetDevice(ctx->device));
    if ((hipStream_t)producer_stream == ctx->stream) return 0;
    HIP_TRY(hipEventRecord(ctx->order_ev, (hipStream_t)producer_stream));
    HIP_TRY(hipStreamWaitEvent(ctx->stream, ctx->order_ev, 0));
    return 0;
}

extern "C" int ssa_ctx_sync(ssa_ctx *ctx) {
    if (!ctx) return SSA_ERR_ARG;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return 0;
}

extern "C" int ssa_ctx_enable_timing(ssa_ctx *ctx, int on) {
    if (!ctx) return SSA_ERR_ARG;
    ctx->timing = on != 0;
    return 0;
}

extern "C" int ssa_ctx_read_timing(ssa_ctx *ctx, const char *kernel, double *avg_ms, uint64_t *launches) {
    if (!ctx || !kernel) return SSA_ERR_ARG;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    double total = 0;
    uint64_t cnt = 0;
    // the context's own launches and, after calls of more than one slice, its twin's (whose launches overlap the
    // context's: their event times are then not exclusive -- DESIGN.md section 5)
    for (ssa_ctx *c : {ctx, ctx->twin}) {
        if (!c) continue;
        if (c != ctx) HIP_TRY(hipStreamSynchronize(c->stream));
        auto it = c->timed.find(kernel);
        if (it == c->timed.end()) continue;
        for (auto &t : it->second) {
            float ms = 0;
            if (hipEventElapsedTime(&ms, t.start, t.stop) == hipSuccess) {
                total += ms;
                cnt++;
            }
            (void)hipEventDestroy(t.start);
            (void)hipEventDestroy(t.stop);
        }
        c->timed.erase(it);
    }
    if (avg_ms) *avg_ms = cnt ? total / (double)cnt : 0.0;
    if (launches) *launches = cnt;
    return 0;
}

// ------------------------------------------------------------------ device entry points
extern "C" int ssa_hash_message_many_device(ssa_ctx *ctx, const uint8_t *d_sigs, const uint8_t *d_pks,
                                            const uint8_t *d_msgs, const uint64_t *d_msg_off,
                                            size_t msg_stride, size_t msg_len, size_t n,
                                            uint8_t *d_digests_out) {
    if (!ctx || (n && (!d_sigs || !d_pks || !d_digests_out))) return SSA_ERR_ARG;
    if (int rc = check_msgs(d_msgs, d_msg_off, msg_stride, msg_len, n)) return rc;
    if (n == 0) return 0;
    HIP_TRY(hipSetDevice(ctx->device));
    MsgView mv{d_msgs, d_msg_off, msg_stride, msg_len};
    return timed_launch(ctx, "ssa_k_hash", [&] {
        hipLaunchKernelGGL(ssa_k_hash, dim3(grid_for(n, 256)), dim3(256), 0, ctx->stream, ctx->d_params,
                           d_sigs, d_pks, mv, n, (u64 *)nullptr, d_digests_out, (const u32 *)nullptr, 0u);
    });
}

extern "C" int ssa_rescue_hash_many_device(ssa_ctx *ctx, const uint64_t *d_felts, uint32_t felts_per_row,
                                           size_t n, uint64_t *d_digests_out) {
    if (!ctx || (n && (!d_digests_out || (felts_per_row && !d_felts)))) return SSA_ERR_ARG;
    if (n == 0) return 0;
    HIP_TRY(hipSetDevice(ctx->device));
    return timed_launch(ctx, "ssa_k_rescue", [&] {
        hipLaunchKernelGGL(ssa_k_rescue, dim3(grid_for(n, 256)), dim3(256), 0, ctx->stream, ctx->d_params,
                           (const u64 *)d_felts, felts_per_row, n, (u64 *)d_digests_out);
    });
}

// one chunk of challenge hashes on `hs` (the shared upload pipeline of the host-buffer entry points)
int ssa_internal_hash_chunk(ssa_ctx *ctx, hipStream_t hs, const uint8_t *d_sigs, const uint8_t *d_pks, const uint8_t *d_msgs,
                            const uint64_t *d_off, size_t msg_stride, size_t msg_len, size_t cnt, uint64_t *d_h) {
    MsgView mv{d_msgs, d_off, msg_stride, msg_len};
    hipLaunchKernelGGL(ssa_k_hash, dim3(grid_for(cnt, 256)), dim3(256), 0, hs, ctx->d_params, d_sigs, d_pks, mv, cnt,
                       (u64 *)d_h, (u8 *)nullptr, (const u32 *)nullptr, 0u);
    HIP_TRY(hipGetLastError());
    return 0;
}

int ssa_internal_hash_scalars(ssa_ctx *ctx, const uint8_t *d_sigs, const uint8_t *d_pks, const uint8_t *d_msgs,
                              const uint64_t *d_msg_off, size_t msg_stride, size_t msg_len, size_t n) {
    if (ctx->ws_h.reserve(n * 4 * sizeof(u64))) return SSA_ERR_HIP;
    MsgView mv{d_msgs, d_msg_off, msg_stride, msg_len};
    return timed_launch(ctx, "ssa_k_hash", [&] {
        hipLaunchKernelGGL(ssa_k_hash, dim3(grid_for(n, 256)), dim3(256), 0, ctx->stream, ctx->d_params,
                           d_sigs, d_pks, mv, n, (u64 *)ctx->ws_h.p, (u8 *)nullptr, (const u32 *)nullptr, 0u);
    });
}

static int verify_launch(ssa_ctx *ctx, const uint8_t *d_sigs, const uint8_t *d_pks, const uint8_t *d_pk_inf,
                         const uint8_t *d_msgs, const uint64_t *d_msg_off, size_t msg_stride, size_t msg_len, size_t n,
                         uint32_t flags, uint8_t *d_status_out, unsigned long long *d_fail);

extern "C" int ssa_verify_many_device(ssa_ctx *ctx, const uint8_t *d_sigs, const uint8_t *d_pks,
                                      const uint8_t *d_pk_inf, const uint8_t *d_msgs,
                                      const uint64_t *d_msg_off, size_t msg_stride, size_t msg_len,
                                      size_t n, uint32_t flags, uint8_t *d_status_out,
                                      uint64_t *d_n_fail_out) {
    if (!ctx || (n && (!d_sigs || !d_pks || !d_status_out))) return SSA_ERR_ARG;
    if (int rc = check_msgs(d_msgs, d_msg_off, msg_stride, msg_len, n)) return rc;
    HIP_TRY(hipSetDevice(ctx->device));
    unsigned long long *d_fail = d_n_fail_out ? (unsigned long long *)d_n_fail_out
                                              : (unsigned long long *)ctx->ws_fail.p;
    HIP_TRY(hipMemsetAsync(d_fail, 0, sizeof(unsigned long long), ctx->stream));
    if (n == 0) return 0;
    return verify_launch(ctx, d_sigs, d_pks, d_pk_inf, d_msgs, d_msg_off, msg_stride, msg_len, n, flags, d_status_out,
                         d_fail);
}

// message view of the lanes from `lo` on (offsets are absolute into msgs: only the offset table moves)
static inline MsgView msg_slice(const MsgView &mv, size_t lo) {
    MsgView s = mv;
    if (mv.off) s.off = mv.off + lo;
    else if (mv.msgs) s.msgs = mv.msgs + lo * mv.stride;
    return s;
}

// The end game of an ssa_k_verify launch over cnt lanes: which groups run in pieces, where their work is cut.  The cuts
// follow instruction counts (a doubling 2642 VALU instructions, a mixed addition 3738; the table build ~57 k in front of
// the first pass, the comb for G and the comparison ~45 k behind the last): the first piece is half of a lane's work, the
// next a quarter, ... the last two equal (SSA_TAIL_UNIFORM=1: equal pieces); a piece never spans the two passes of
// SSA_FLAG_CHECK_TORSION.
struct TailKnobs {
    unsigned pieces, gens, waves, block, min_main;
    bool uniform, reversed;
};
static TailPlan tail_plan_of(const TailKnobs &kn, size_t cnt, uint32_t flags);
static TailPlan tail_plan(const ssa_ctx *ctx, size_t cnt, uint32_t flags) {
    return tail_plan_of({ctx->tail_pieces, ctx->tail_gens, ctx->verify_waves, ctx->verify_block, ctx->tail_min_main,
                         ctx->tail_uniform, ctx->tail_reversed}, cnt, flags);
}
static TailPlan tail_plan_of(const TailKnobs &kn, size_t cnt, uint32_t flags) {
    const TailKnobs *ctx = &kn;
    TailPlan tp{};
    const bool torsion = (flags & SSA_FLAG_CHECK_TORSION) != 0;
    tp.whole[0] = 0u | 2u | 4u | ((u32)LADDER_STEPS_Q << 16);
    tp.whole[1] = 1u | 2u | 4u | ((u32)LADDER_STEPS << 16);
    const u32 n_groups = (u32)((cnt + 63) / 64);
    const unsigned cap = torsion ? (unsigned)VP_MAX - 1u : (unsigned)VP_MAX;      // (the pass boundary is one more cut)
    const unsigned want = ctx->pieces < cap ? ctx->pieces : cap;
    const u32 tail0 = ctx->gens * ctx->waves;
    if (want < 2 || ctx->block != 256 || tail0 == 0 || n_groups < tail0 + ctx->min_main * ctx->waves) return tp;
    const u32 main_groups = ((n_groups - tail0) / 4u) * 4u;
    tp.main_blocks = main_groups / 4u;
    tp.tail_groups = ((n_groups - main_groups + 3u) / 4u) * 4u;
    // the lane's work as one sequence of steps: [pass 0 windows] [pass 1 windows], with the table build in front and the
    // comb + comparison behind
    const double DBL = 2642.0, ADD = 3738.0, TABLE = 57000.0, TAIL = 45000.0;
    struct Step { int pass, it; double cost; };
    std::vector<Step> steps;
    for (int pass = torsion ? 0 : 1; pass < 2; pass++)
        for (int it = 0; it < (pass == 0 ? LADDER_STEPS_Q : LADDER_STEPS); it++)
            steps.push_back({pass, it, (pass == 0 ? (double)QNAF_GAP[it + 1] : 5.0) * DBL + ADD});
    double total = TABLE + TAIL;
    for (const Step &st : steps) total += st.cost;
    // cumulative targets of the pieces
    std::vector<double> target;
    double frac = 0.0, f = 0.5;
    for (unsigned k = 0; k + 1 < want; k++) {
        frac += ctx->uniform ? 1.0 / (double)want : f;
        if (k + 2 < want) f *= 0.5;
        target.push_back(total * frac);
    }
    // cut in front of the step that would cross a target (and between the passes, always)
    std::vector<std::pair<size_t, size_t>> pieces;
    size_t cur = 0, tix = 0;
    double acc = TABLE;
    for (size_t idx = 0; idx < steps.size(); idx++) {
        bool cut = false;
        if (idx > cur) {
            if (steps[idx].pass != steps[idx - 1].pass) cut = true;
            else if (tix < target.size() && acc + steps[idx].cost * 0.5 >= target[tix]) cut = true;
        }
        if (cut) {
            pieces.push_back({cur, idx});
            cur = idx;
        }
        while (tix < target.size() && acc + steps[idx].cost * 0.5 >= target[tix]) tix++;     // targets reached
        acc += steps[idx].cost;
    }
    pieces.push_back({cur, steps.size()});
    if (pieces.size() < 2 || pieces.size() > (size_t)VP_MAX) return tp;        // (n_pieces stays 0: no end game)
    for (const auto &pc : pieces) {
        const int pass = steps[pc.first].pass, lo = steps[pc.first].it, hi = steps[pc.second - 1].it + 1;
        const int n_steps = pass == 0 ? LADDER_STEPS_Q : LADDER_STEPS;
        tp.ph[tp.n_pieces++] = (u32)pass | (lo == 0 ? 2u : 0u) | (hi == n_steps ? 4u : 0u) | ((u32)lo << 8) | ((u32)hi << 16);
    }
    tp.reversed = ctx->reversed ? 1u : 0u;
    return tp;
}

// the plan for explicit knobs: no context and no device needed, so the host logic is tested on the CPU
// (tests/test_tail_plan.py: the pieces cover every window of every pass exactly once and never span two passes, the
// grid is what the kernel assumes).  out: n_pieces, tail_groups, main_blocks, grid blocks, ph[0..7], whole[0..1].
extern "C" int ssa_debug_tail_plan(unsigned waves, unsigned pieces, unsigned gens, int uniform, unsigned min_main, size_t n,
                                   uint32_t flags, uint32_t out[14]) {
    if (!out) return SSA_ERR_ARG;
    const TailPlan tp = tail_plan_of({pieces, gens, waves, 256u, min_main, uniform != 0, false}, n, flags);
    out[0] = tp.n_pieces;
    out[1] = tp.tail_groups;
    out[2] = tp.main_blocks;
    out[3] = tp.n_pieces ? tail_grid_blocks(tp) : (uint32_t)((n + 255) / 256);
    for (int k = 0; k < VP_MAX; k++) out[4 + k] = tp.ph[k];
    out[12] = tp.whole[0];
    out[13] = tp.whole[1];
    return 0;
}

// ssa_k_verify over n lanes whose challenge scalars are in d_h, in slices of at most ctx->lane_slice lanes: the 4 KB
// per-lane table workspace never exceeds one slice (the caller has reserved it).  *d_fail is added to.
static int verify_slices(ssa_ctx *ctx, const uint8_t *d_sigs, const uint8_t *d_pks, const uint8_t *d_pk_inf,
                         const u64 *d_h, size_t n, uint32_t flags, uint8_t *d_status_out, unsigned long long *d_fail) {
    const size_t slice = ctx->lane_slice < n ? ctx->lane_slice : n;
    for (size_t lo = 0; lo < n; lo += slice) {
        const size_t cnt = n - lo < slice ? n - lo : slice;
        const TailPlan tp = tail_plan(ctx, cnt, flags);
        unsigned blocks = grid_for(cnt, ctx->verify_block);
        if (tp.n_pieces) {
            if (ctx->tail_done.reserve(2 * (size_t)tp.tail_groups * sizeof(u32)) ||       // finished pieces, claimed pieces
                ctx->tail_park.reserve((size_t)tp.tail_groups * PARK_WORDS * 64 * sizeof(u64)))
                return SSA_ERR_HIP;
            HIP_TRY(hipMemsetAsync(ctx->tail_done.p, 0, 2 * (size_t)tp.tail_groups * sizeof(u32), ctx->stream));
            blocks = tail_grid_blocks(tp);
        }
        int rc = timed_launch(ctx, "ssa_k_verify", [&] {
            hipLaunchKernelGGL(ssa_k_verify, dim3(blocks), dim3(ctx->verify_block), 0,
                               ctx->stream, d_sigs + 81 * lo, d_pks + 96 * lo, d_pk_inf ? d_pk_inf + lo : nullptr,
                               d_h + 4 * lo, (const u64 *)ctx->d_gtab, (u64 *)ctx->ws_tab.p, cnt, flags,
                               d_status_out + lo, d_fail, tp, (u32 *)ctx->tail_done.p, (u64 *)ctx->tail_park.p);
        });
        if (rc) return rc;
    }
    return 0;
}

// hash + verification of ONE slice (cnt <= c->lane_slice lanes) on c->stream with c's workspaces; *d_fail is added to
static int verify_one_slice(ssa_ctx *c, const uint8_t *d_sigs, const uint8_t *d_pks, const uint8_t *d_pk_inf,
                            const MsgView &mv, size_t cnt, uint32_t flags, uint8_t *d_status_out,
                            unsigned long long *d_fail) {
    if (c->ws_h.reserve(cnt * 4 * sizeof(u64))) return SSA_ERR_HIP;
    if (c->ws_tab.reserve(cnt * (size_t)(PTAB_ENTRIES * PTAB_ENTRY_U64) * sizeof(u64))) return SSA_ERR_HIP;
    int rc = timed_launch(c, "ssa_k_hash", [&] {
        hipLaunchKernelGGL(ssa_k_hash, dim3(grid_for(cnt, 256)), dim3(256), 0, c->stream, c->d_params, d_sigs, d_pks, mv,
                           cnt, (u64 *)c->ws_h.p, (u8 *)nullptr, (const u32 *)nullptr, 0u);
    });
    if (rc) return rc;
    return verify_slices(c, d_sigs, d_pks, d_pk_inf, (const u64 *)c->ws_h.p, cnt, flags, d_status_out, d_fail);
}

// the kernels of one verification batch on ctx->stream; *d_fail is added to, not reset
static int verify_launch(ssa_ctx *ctx, const uint8_t *d_sigs, const uint8_t *d_pks, const uint8_t *d_pk_inf,
                         const uint8_t *d_msgs, const uint64_t *d_msg_off, size_t msg_stride, size_t msg_len, size_t n,
                         uint32_t flags, uint8_t *d_status_out, unsigned long long *d_fail) {
    MsgView mv{d_msgs, d_msg_off, msg_stride, msg_len};
    // small batches: one wave per signature (low latency); large ones: one lane per signature (throughput)
    const size_t coop_lim = (flags & SSA_FLAG_CHECK_TORSION) ? ctx->coop_max_n_torsion : ctx->coop_max_n;
    const bool coop = (flags & SSA_FLAG_FORCE_COOP) || (!(flags & SSA_FLAG_FORCE_LANE) && n <= coop_lim);
    if (coop) {
        return timed_launch(ctx, "ssa_k_verify_coop", [&] {
            hipLaunchKernelGGL(ssa_k_verify_coop, dim3((unsigned)n), dim3(128), 0, ctx->stream, ctx->d_params, d_sigs,
                               d_pks, d_pk_inf, mv, (const u64 *)ctx->d_gtab, n, flags, d_status_out, d_fail);
        });
    }
    // The per-lane workspaces (32 B of challenge scalar, 4 KB of table) are sized for ONE slice of at most
    // ctx->lane_slice lanes, whatever n is (4.3 GB of tables at the default 2^20; the reference takes slices of any
    // length, src/batch.rs:31-50): a larger batch runs slice after slice, into the caller's one status array and the one
    // rejection counter.  At n <= lane_slice this is the single pair of launches it always was.
    const size_t slice = ctx->lane_slice < n ? ctx->lane_slice : n;
    if (n <= slice) return verify_one_slice(ctx, d_sigs, d_pks, d_pk_inf, mv, n, flags, d_status_out, d_fail);
    // More than one slice: the slices alternate between the context's stream and its twin's (a second set of
    // workspaces), so that one slice's kernels fill the tails of the other's -- ordered after everything queued on
    // ctx->stream before the call, and ctx->stream continues after both.
    ssa_ctx *tw = ssa_internal_twin(ctx);
    if (tw) {
        HIP_TRY(hipEventRecord(ctx->order_ev, ctx->stream));
        HIP_TRY(hipStreamWaitEvent(tw->stream, ctx->order_ev, 0));
    }
    int rc = 0;
    size_t j = 0;
    for (size_t lo = 0; lo < n && rc == 0; lo += slice, j++) {
        const size_t cnt = n - lo < slice ? n - lo : slice;
        ssa_ctx *c = (tw && (j & 1u)) ? tw : ctx;
        rc = verify_one_slice(c, d_sigs + 81 * lo, d_pks + 96 * lo, d_pk_inf ? d_pk_inf + lo : nullptr, msg_slice(mv, lo), cnt,
                              flags, d_status_out + lo, d_fail);
    }
    if (tw) {     // (also on an error: whatever was queued on the twin's stream is still ordered before the caller's next step)
        if (hipEventRecord(tw->order_ev, tw->stream) != hipSuccess ||
            hipStreamWaitEvent(ctx->stream, tw->order_ev, 0) != hipSuccess)
            return rc ? rc : SSA_ERR_HIP;
    }
    return rc;
}

// the throughput (variable-time) signer's launch; arguments checked by the caller (ssa_sign.hip)
int ssa_internal_sign_vartime(ssa_ctx *ctx, const uint8_t *d_sks, const uint8_t *d_nonces, const uint8_t *d_msgs,
                              const uint64_t *d_msg_off, size_t msg_stride, size_t msg_len, size_t n, uint8_t *d_pks_out,
                              uint8_t *d_sigs_out) {
    MsgView mv{d_msgs, d_msg_off, msg_stride, msg_len};
    return timed_launch(ctx, "ssa_k_sign", [&] {
        hipLaunchKernelGGL(ssa_k_sign, dim3(grid_for(n, 256)), dim3(256), 0, ctx->stream, ctx->d_params,
                           (const u64 *)ctx->d_gtab, d_sks, d_nonces, mv, n, d_pks_out, d_sigs_out);
    });
}

extern "C" int ssa_keygen_sign_many_device(ssa_ctx *ctx, const uint8_t *d_sks, const uint8_t *d_nonces,
                                           const uint8_t *d_msgs, const uint64_t *d_msg_off,
                                           size_t msg_stride, size_t msg_len, size_t n,
                                           uint8_t *d_pks_out, uint8_t *d_sigs_out) {
    return ssa_keygen_sign_many_ex_device(ctx, d_sks, d_nonces, d_msgs, d_msg_off, msg_stride, msg_len, n, 0u, d_pks_out,
                                          d_sigs_out);
}

extern "C" int ssa_decompress_many_device(ssa_ctx *ctx, const uint8_t *d_compressed, size_t n,
                                          uint8_t *d_pks_out, uint8_t *d_pk_inf_out, uint8_t *d_status_out) {
    if (!ctx || (n && (!d_compressed || !d_pks_out || !d_status_out)) || n > SSA_MAX_BATCH) return SSA_ERR_ARG;
    if (n == 0) return 0;
    HIP_TRY(hipSetDevice(ctx->device));
    return timed_launch(ctx, "ssa_k_decompress", [&] {
        hipLaunchKernelGGL(ssa_k_decompress, dim3(grid_for(n, 256)), dim3(256), 0, ctx->stream, d_compressed, n,
                           d_pks_out, d_pk_inf_out, d_status_out);
    });
}

// ------------------------------------------------------------------ host entry points
// Large host-buffer batch: the shared upload + hash pipeline (ssa_ctx.hpp: pipelined_upload_hash), then ONE
// verification launch over the whole batch: the ladder kernel keeps its full-size grid, only the first chunk's
// upload is exposed.
static int verify_many_pipelined(ssa_ctx *ctx, const uint8_t *sigs, const uint8_t *pks, const uint8_t *pk_inf,
                                 const uint8_t *msgs, const uint64_t *msg_off, size_t msg_stride, size_t msg_len,
                                 size_t n, uint32_t flags, uint8_t *status_out, uint64_t *n_fail_out, bool *used) {
    PipelinedInputs pin;      // its destructor drains the side streams on every error return below
    *used = false;
    if (ctx->pin_out.reserve(n)) return 0;          // (no page-locked memory: the staged path)
    if (int rc = pipelined_upload_hash(ctx, sigs, pks, pk_inf, msgs, msg_off, msg_stride, msg_len, n, pin, used)) return rc;
    if (!*used) return 0;
    const size_t slice = ctx->lane_slice < n ? ctx->lane_slice : n;
    if (ctx->st_status.reserve(n + 16) ||
        ctx->ws_tab.reserve(slice * (size_t)(PTAB_ENTRIES * PTAB_ENTRY_U64) * sizeof(u64)))
        return SSA_ERR_HIP;
    unsigned long long *d_fail = (unsigned long long *)ctx->ws_fail.p;
    HIP_TRY(hipMemsetAsync(d_fail, 0, sizeof(unsigned long long), ctx->stream));
    // (the pipeline hashed the whole batch into ws_h, 32 B per lane, while it was uploading)
    if (int rc = verify_slices(ctx, pin.s.sigs, pin.s.pks, pin.s.inf, (const u64 *)ctx->ws_h.p, n, flags,
                               (u8 *)ctx->st_status.p, d_fail))
        return rc;
    HIP_TRY(hipMemcpyAsync(ctx->pin_out.p, ctx->st_status.p, n, hipMemcpyDeviceToHost, ctx->stream));
    unsigned long long nf = 0;
    HIP_TRY(hipMemcpyAsync(&nf, d_fail, sizeof nf, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    pin.done();
    std::memcpy(status_out, ctx->pin_out.p, n);
    if (n_fail_out) *n_fail_out = nf;
    return 0;
}

static int verify_many_host_one(ssa_ctx *ctx, const uint8_t *sigs, const uint8_t *pks, const uint8_t *pk_inf,
                                const uint8_t *msgs, const uint64_t *msg_off, size_t msg_stride, size_t msg_len, size_t n,
                                uint32_t flags, uint8_t *status_out, uint64_t *n_fail_out);

extern "C" int ssa_verify_many(ssa_ctx *ctx, const uint8_t *sigs, const uint8_t *pks, const uint8_t *pk_inf,
                               const uint8_t *msgs, const uint64_t *msg_off, size_t msg_stride,
                               size_t msg_len, size_t n, uint32_t flags, uint8_t *status_out,
                               uint64_t *n_fail_out) {
    if (!ctx || (n && (!sigs || !pks || !status_out))) return SSA_ERR_ARG;
    if (int rc = check_msgs(msgs, msg_off, msg_stride, msg_len, n)) return rc;
    if (msg_off)
        for (size_t i = 0; i < n; i++)
            if (msg_off[i + 1] < msg_off[i] || msg_off[i + 1] - msg_off[i] > 0xffffffffull) return SSA_ERR_ARG;
    if (n_fail_out) *n_fail_out = 0;
    if (n == 0) return 0;
    HIP_TRY(hipSetDevice(ctx->device));
    if (n <= ctx->lane_slice)
        return verify_many_host_one(ctx, sigs, pks, pk_inf, msgs, msg_off, msg_stride, msg_len, n, flags, status_out,
                                    n_fail_out);
    std::mutex mu;
    uint64_t total = 0;
    const int rc = run_host_slices(ctx, n, ctx->lane_slice, [&](ssa_ctx *c, size_t lo, size_t cnt) {
        const HostMsgSlice ms(msgs, msg_off, msg_stride, lo, cnt);
        uint64_t nf = 0;
        const int r = verify_many_host_one(c, sigs + 81 * lo, pks + 96 * lo, pk_inf ? pk_inf + lo : nullptr, ms.msgs, ms.offp,
                                           msg_stride, msg_len, cnt, flags, status_out + lo, &nf);
        std::lock_guard<std::mutex> lock(mu);
        total += nf;
        return r;
    });
    if (rc) return rc;
    if (n_fail_out) *n_fail_out = total;
    return 0;
}

// one slice (n <= ctx->lane_slice, or a batch for the cooperative kernel) from host buffers
static int verify_many_host_one(ssa_ctx *ctx, const uint8_t *sigs, const uint8_t *pks, const uint8_t *pk_inf,
                                const uint8_t *msgs, const uint64_t *msg_off, size_t msg_stride, size_t msg_len, size_t n,
                                uint32_t flags, uint8_t *status_out, uint64_t *n_fail_out) {
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t coop_lim = (flags & SSA_FLAG_CHECK_TORSION) ? ctx->coop_max_n_torsion : ctx->coop_max_n;
    const bool lane_kernels = !(flags & SSA_FLAG_FORCE_COOP) && ((flags & SSA_FLAG_FORCE_LANE) || n > coop_lim);
    if (lane_kernels && n >= ctx->pipeline_min_n && ctx->pipeline_chunks > 1) {
        bool used = false;
        const int rc = verify_many_pipelined(ctx, sigs, pks, pk_inf, msgs, msg_off, msg_stride, msg_len, n, flags,
                                             status_out, n_fail_out, &used);
        if (used) return rc;
    }
    StagedInputs s;
    const void *p;
    if (int rc = stage_up(ctx, ctx->st_sigs, sigs, n * 81, &p)) return rc;
    s.sigs = (const u8 *)p;
    if (int rc = stage_up(ctx, ctx->st_pks, pks, n * 96, &p)) return rc;
    s.pks = (const u8 *)p;
    if (pk_inf) {
        if (int rc = stage_up(ctx, ctx->st_inf, pk_inf, n, &p)) return rc;
        s.inf = (const u8 *)p;
    }
    if (int rc = stage_msgs(ctx, msgs, msg_off, msg_stride, msg_len, n, s)) return rc;
    if (ctx->st_status.reserve(n + 16)) return SSA_ERR_HIP;
    unsigned long long *d_fail = (unsigned long long *)ctx->ws_fail.p;
    if (int rc = ssa_verify_many_device(ctx, s.sigs, s.pks, s.inf, s.msgs, s.off, msg_stride, msg_len, n,
                                        flags, (u8 *)ctx->st_status.p, (uint64_t *)d_fail))
        return rc;
    HIP_TRY(hipMemcpyAsync(status_out, ctx->st_status.p, n, hipMemcpyDeviceToHost, ctx->stream));
    unsigned long long nf = 0;
    HIP_TRY(hipMemcpyAsync(&nf, d_fail, sizeof nf, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (n_fail_out) *n_fail_out = nf;
    return 0;
}

extern "C" int ssa_verify_batch(ssa_ctx *ctx, const uint8_t *sigs, const uint8_t *pks, const uint8_t *pk_inf,
                                const uint8_t *msgs, const uint64_t *msg_off, size_t msg_stride, size_t msg_len,
                                size_t n, uint32_t flags) {
    if (n == 0) return ctx ? SSA_OK : SSA_ERR_ARG;  // empty batch verifies (src/batch.rs)
    std::vector<uint8_t> status(n);
    uint64_t nf = 0;
    // the reference's verify_batch decompresses R with its flag byte (src/batch.rs:104)
    if (int rc = ssa_verify_many(ctx, sigs, pks, pk_inf, msgs, msg_off, msg_stride, msg_len, n,
                                 flags | SSA_FLAG_SIG_FLAG_BYTE, status.data(), &nf))
        return rc;
    if (nf == 0) return SSA_OK;
    // the reference panics on an undecodable input before it compares anything (src/batch.rs:67,104): SSA_MALFORMED
    // dominates; then the subgroup check of SSA_FLAG_CHECK_TORSION (an extension: src/batch.rs has none)
    bool any_pk = false;
    for (uint8_t s : status) {
        if (s == SSA_MALFORMED) return SSA_MALFORMED;
        any_pk = any_pk || s == SSA_INVALID_PUBLIC_KEY;
    }
    return any_pk ? SSA_INVALID_PUBLIC_KEY : SSA_INVALID_SIGNATURE;
}

extern "C" int ssa_verify(ssa_ctx *ctx, const uint8_t sig[SSA_SIGNATURE_LENGTH],
                          const uint8_t pk[SSA_AFFINE_PK_LENGTH], const uint8_t *msg, size_t msg_len,
                          uint32_t flags) {
    uint8_t st = SSA_MALFORMED;
    if (int rc = ssa_verify_many(ctx, sig, pk, nullptr, msg, nullptr, msg_len, msg_len, 1, flags, &st, nullptr))
        return rc;
    return st;
}

extern "C" int ssa_hash_message_many(ssa_ctx *ctx, const uint8_t *sigs, const uint8_t *pks, const uint8_t *msgs,
                                     const uint64_t *msg_off, size_t msg_stride, size_t msg_len, size_t n,
                                     uint8_t *digests_out) {
    if (!ctx || (n && (!sigs || !pks || !digests_out))) return SSA_ERR_ARG;
    if (int rc = check_msgs(msgs, msg_off, msg_stride, msg_len, n)) return rc;
    if (n == 0) return 0;
    HIP_TRY(hipSetDevice(ctx->device));
    StagedInputs s;
    const void *p;
    if (int rc = stage_up(ctx, ctx->st_sigs, sigs, n * 81, &p)) return rc;
    s.sigs = (const u8 *)p;
    if (int rc = stage_up(ctx, ctx->st_pks, pks, n * 96, &p)) return rc;
    s.pks = (const u8 *)p;
    if (int rc = stage_msgs(ctx, msgs, msg_off, msg_stride, msg_len, n, s)) return rc;
    if (ctx->st_aux.reserve(n * 32)) return SSA_ERR_HIP;
    if (int rc = ssa_hash_message_many_device(ctx, s.sigs, s.pks, s.msgs, s.off, msg_stride, msg_len, n,
                                              (u8 *)ctx->st_aux.p))
        return rc;
    HIP_TRY(hipMemcpyAsync(digests_out, ctx->st_aux.p, n * 32, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return 0;
}

extern "C" int ssa_rescue_hash_many(ssa_ctx *ctx, const uint64_t *felts, uint32_t felts_per_row, size_t n,
                                    uint64_t *digests_out) {
    if (!ctx || (n && (!digests_out || (felts_per_row && !felts)))) return SSA_ERR_ARG;
    if (n == 0) return 0;
    HIP_TRY(hipSetDevice(ctx->device));
    const void *p;
    if (int rc = stage_up(ctx, ctx->st_aux, felts, n * (size_t)felts_per_row * 8, &p)) return rc;
    if (ctx->st_aux2.reserve(n * 32)) return SSA_ERR_HIP;
    if (int rc = ssa_rescue_hash_many_device(ctx, (const uint64_t *)p, felts_per_row, n,
                                             (uint64_t *)ctx->st_aux2.p))
        return rc;
    HIP_TRY(hipMemcpyAsync(digests_out, ctx->st_aux2.p, n * 32, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return 0;
}

extern "C" int ssa_keygen_sign_many(ssa_ctx *ctx, const uint8_t *sks, const uint8_t *nonces,
                                    const uint8_t *msgs, const uint64_t *msg_off, size_t msg_stride,
                                    size_t msg_len, size_t n, uint8_t *pks_out, uint8_t *sigs_out) {
    if (n && !pks_out) return SSA_ERR_ARG;
    return ssa_keygen_sign_many_ex(ctx, sks, nonces, msgs, msg_off, msg_stride, msg_len, n, 0u, pks_out, sigs_out);
}

extern "C" int ssa_decompress_many(ssa_ctx *ctx, const uint8_t *compressed, size_t n, uint8_t *pks_out,
                                   uint8_t *pk_inf_out, uint8_t *status_out) {
    if (!ctx || (n && (!compressed || !pks_out || !status_out))) return SSA_ERR_ARG;
    if (n == 0) return 0;
    HIP_TRY(hipSetDevice(ctx->device));
    const void *p;
    if (int rc = stage_up(ctx, ctx->st_sigs, compressed, n * 49, &p)) return rc;
    if (ctx->st_aux.reserve(n * 96) || ctx->st_aux2.reserve(n + 16) || ctx->st_status.reserve(n + 16))
        return SSA_ERR_HIP;
    if (int rc = ssa_decompress_many_device(ctx, (const u8 *)p, n, (u8 *)ctx->st_aux.p, (u8 *)ctx->st_aux2.p,
                                            (u8 *)ctx->st_status.p))
        return rc;
    HIP_TRY(hipMemcpyAsync(pks_out, ctx->st_aux.p, n * 96, hipMemcpyDeviceToHost, ctx->stream));
    if (pk_inf_out) HIP_TRY(hipMemcpyAsync(pk_inf_out, ctx->st_aux2.p, n, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(status_out, ctx->st_status.p, n, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return 0;
}

static int verify_keyed_host_one(ssa_ctx *ctx, const uint8_t *keyed, const uint8_t *msgs, const uint64_t *msg_off,
                                 size_t msg_stride, size_t msg_len, size_t n, uint32_t flags, uint8_t *status_out,
                                 uint64_t *n_fail_out);

extern "C" int ssa_verify_keyed_many(ssa_ctx *ctx, const uint8_t *keyed, const uint8_t *msgs,
                                     const uint64_t *msg_off, size_t msg_stride, size_t msg_len, size_t n,
                                     uint32_t flags, uint8_t *status_out, uint64_t *n_fail_out) {
    if (!ctx || (n && (!keyed || !status_out))) return SSA_ERR_ARG;
    if (int rc = check_msgs(msgs, msg_off, msg_stride, msg_len, n)) return rc;
    if (msg_off)
        for (size_t i = 0; i < n; i++)
            if (msg_off[i + 1] < msg_off[i] || msg_off[i + 1] - msg_off[i] > 0xffffffffull) return SSA_ERR_ARG;
    if (n_fail_out) *n_fail_out = 0;
    if (n == 0) return 0;
    HIP_TRY(hipSetDevice(ctx->device));
    if (n <= ctx->lane_slice)
        return verify_keyed_host_one(ctx, keyed, msgs, msg_off, msg_stride, msg_len, n, flags, status_out, n_fail_out);
    std::mutex mu;
    uint64_t total = 0;
    const int rc = run_host_slices(ctx, n, ctx->lane_slice, [&](ssa_ctx *c, size_t lo, size_t cnt) {
        const HostMsgSlice ms(msgs, msg_off, msg_stride, lo, cnt);
        uint64_t nf = 0;
        const int r = verify_keyed_host_one(c, keyed + 130 * lo, ms.msgs, ms.offp, msg_stride, msg_len, cnt, flags,
                                            status_out + lo, &nf);
        std::lock_guard<std::mutex> lock(mu);
        total += nf;
        return r;
    });
    if (rc) return rc;
    if (n_fail_out) *n_fail_out = total;
    return 0;
}

// one slice of KeyedSignature records (n <= ctx->lane_slice) from host buffers
static int verify_keyed_host_one(ssa_ctx *ctx, const uint8_t *keyed, const uint8_t *msgs, const uint64_t *msg_off,
                                 size_t msg_stride, size_t msg_len, size_t n, uint32_t flags, uint8_t *status_out,
                                 uint64_t *n_fail_out) {
    HIP_TRY(hipSetDevice(ctx->device));
    StagedInputs s;
    const void *p;
    if (int rc = stage_up(ctx, ctx->st_coeffs, keyed, n * 130, &p)) return rc;
    if (ctx->st_sigs.reserve(n * 81) || ctx->st_pks.reserve(n * 96) || ctx->st_inf.reserve(n + 16) ||
        ctx->st_status.reserve(n + 16))
        return SSA_ERR_HIP;
    hipLaunchKernelGGL(ssa_k_unpack_keyed, dim3(grid_for(n, 256)), dim3(256), 0, ctx->stream, (const u8 *)p, n,
                       (u8 *)ctx->st_pks.p, (u8 *)ctx->st_inf.p, (u8 *)ctx->st_sigs.p);
    HIP_TRY(hipGetLastError());
    if (int rc = stage_msgs(ctx, msgs, msg_off, msg_stride, msg_len, n, s)) return rc;
    unsigned long long *d_fail = (unsigned long long *)ctx->ws_fail.p;
    if (int rc = ssa_verify_many_device(ctx, (const u8 *)ctx->st_sigs.p, (const u8 *)ctx->st_pks.p,
                                        (const u8 *)ctx->st_inf.p, s.msgs, s.off, msg_stride, msg_len, n, flags,
                                        (u8 *)ctx->st_status.p, (uint64_t *)d_fail))
        return rc;
    HIP_TRY(hipMemcpyAsync(status_out, ctx->st_status.p, n, hipMemcpyDeviceToHost, ctx->stream));
    unsigned long long nf = 0;
    HIP_TRY(hipMemcpyAsync(&nf, d_fail, sizeof nf, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (n_fail_out) *n_fail_out = nf;
    return 0;
}

// ------------------------------------------------------------------ keyed context

extern "C" int ssa_keyset_create_device(ssa_ctx *ctx, const uint8_t *d_pks, const uint8_t *d_pk_inf, size_t m,
                                        uint32_t flags, ssa_keyset **out) {
    if (!ctx || !out || !d_pks || m == 0 || m > 0xffffffffull || flags > SSA_KEYSET_LADDER) return SSA_ERR_ARG;
    *out = nullptr;
    HIP_TRY(hipSetDevice(ctx->device));
    ssa_keyset *ks = new ssa_keyset();
    ks->ctx = ctx;
    ks->m = m;
    if (ks->tab.reserve(m * (size_t)(PTAB_ENTRIES * PTAB_ENTRY_U64) * sizeof(u64)) || ks->status.reserve(m + 16) ||
        ks->pks.reserve(m * 96)) {
        ssa_keyset_destroy(ks);
        return SSA_ERR_HIP;
    }
    // the hash kernel reads the keys (x, y_0) through the index: keep a copy so that the caller's buffer can go
    if (hipMemcpyAsync(ks->pks.p, d_pks, m * 96, hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess) {
        ssa_keyset_destroy(ks);
        return SSA_ERR_HIP;
    }
    int rc = timed_launch(ctx, "ssa_k_keyset_build", [&] {
        hipLaunchKernelGGL(ssa_k_keyset_build, dim3(grid_for(m, 256)), dim3(256), 0, ctx->stream,
                           (const u8 *)ks->pks.p, d_pk_inf, m, (u64 *)ks->tab.p, (u8 *)ks->status.p);
    });
    if (rc != 0 || hipStreamSynchronize(ctx->stream) != hipSuccess) {
        ssa_keyset_destroy(ks);
        return rc ? rc : SSA_ERR_HIP;
    }
    // few keys: a comb table per key (no doublings at verification time); many keys: the ladder tables only
    // (100 MB per key: AUTO takes the combs while they fit the context's HBM budget and 16 GiB, and falls back to the
    // ladder tables when the allocation fails; SSA_KEYSET_COMB insists and reports the failure)
    const size_t comb_bytes = m * KTAB_ENTRIES_PER_KEY * 12 * sizeof(u64);
    const size_t comb_cap = ctx->hbm_budget < ((uint64_t)16 << 30) ? (size_t)ctx->hbm_budget : ((size_t)16 << 30);
    ks->comb = flags == SSA_KEYSET_COMB || (flags == SSA_KEYSET_AUTO && comb_bytes <= comb_cap);
    DevBuf kbase;                                      // 32 x 256 base rows (768 KB) per key, only while the combs are assembled
    if (ks->comb && (ks->ktab.reserve(comb_bytes) || kbase.reserve(m * KBASE_ENTRIES_PER_KEY * 12 * sizeof(u64)))) {
        kbase.release();
        ks->ktab.release();
        (void)hipGetLastError();
        if (flags == SSA_KEYSET_COMB) {
            ssa_keyset_destroy(ks);
            return SSA_ERR_HIP;
        }
        ks->comb = false;
    }
    if (ks->comb) {
        rc = timed_launch(ctx, "ssa_k_keycomb_build", [&] {
            hipLaunchKernelGGL(ssa_k_keycomb_base, dim3(grid_for(m * KBASE_ENTRIES_PER_KEY, 256)), dim3(256), 0,
                               ctx->stream, (const u8 *)ks->pks.p, d_pk_inf, (const u8 *)ks->status.p, m,
                               (u64 *)kbase.p);
            hipLaunchKernelGGL(ssa_k_keycomb_build, dim3(grid_for(m * (KTAB_ENTRIES_PER_KEY / 8), 256)), dim3(256), 0,
                               ctx->stream, (const u64 *)kbase.p, m, (u64 *)ks->ktab.p);
        });
        const bool synced = hipStreamSynchronize(ctx->stream) == hipSuccess;
        kbase.release();
        if (rc != 0 || !synced) {
            ssa_keyset_destroy(ks);
            return rc ? rc : SSA_ERR_HIP;
        }
    }
    ctx->keysets.push_back(ks);
    *out = ks;
    return 0;
}

extern "C" int ssa_keyset_create(ssa_ctx *ctx, const uint8_t *pks, const uint8_t *pk_inf, size_t m, uint32_t flags,
                                 ssa_keyset **out) {
    if (!ctx || !out || !pks || m == 0) return SSA_ERR_ARG;
    HIP_TRY(hipSetDevice(ctx->device));
    const void *p_pks, *p_inf = nullptr;
    if (int rc = stage_up(ctx, ctx->st_pks, pks, m * 96, &p_pks)) return rc;
    if (pk_inf)
        if (int rc = stage_up(ctx, ctx->st_inf, pk_inf, m, &p_inf)) return rc;
    return ssa_keyset_create_device(ctx, (const u8 *)p_pks, (const u8 *)p_inf, m, flags, out);
}

extern "C" void ssa_keyset_destroy(ssa_keyset *ks) {
    if (!ks) return;
    if (ks->ctx) {
        (void)hipSetDevice(ks->ctx->device);
        (void)hipStreamSynchronize(ks->ctx->stream);
        auto &v = ks->ctx->keysets;
        for (size_t i = 0; i < v.size(); i++)
            if (v[i] == ks) {
                v.erase(v.begin() + (long)i);
                break;
            }
        ks->release_all();
    }
    delete ks;
}

extern "C" int ssa_keyset_status(ssa_keyset *ks, uint8_t *status_out) {
    if (!ks || !ks->ctx || !status_out) return SSA_ERR_ARG;
    HIP_TRY(hipSetDevice(ks->ctx->device));
    HIP_TRY(hipMemcpyAsync(status_out, ks->status.p, ks->m, hipMemcpyDeviceToHost, ks->ctx->stream));
    HIP_TRY(hipStreamSynchronize(ks->ctx->stream));
    return 0;
}

extern "C" int ssa_verify_many_indexed_device(ssa_ctx *ctx, ssa_keyset *ks, const uint32_t *d_key_idx,
                                              const uint8_t *d_sigs, const uint8_t *d_msgs, const uint64_t *d_msg_off,
                                              size_t msg_stride, size_t msg_len, size_t n, uint32_t flags,
                                              uint8_t *d_status_out, uint64_t *d_n_fail_out) {
    if (!ctx || !ks || ks->ctx != ctx || (n && (!d_key_idx || !d_sigs || !d_status_out))) return SSA_ERR_ARG;
    if (int rc = check_msgs(d_msgs, d_msg_off, msg_stride, msg_len, n)) return rc;
    HIP_TRY(hipSetDevice(ctx->device));
    unsigned long long *d_fail = d_n_fail_out ? (unsigned long long *)d_n_fail_out
                                              : (unsigned long long *)ctx->ws_fail.p;
    HIP_TRY(hipMemsetAsync(d_fail, 0, sizeof(unsigned long long), ctx->stream));
    if (n == 0) return 0;
    if (ctx->ws_h.reserve(n * 4 * sizeof(u64))) return SSA_ERR_HIP;
    MsgView mv{d_msgs, d_msg_off, msg_stride, msg_len};
    int rc = timed_launch(ctx, "ssa_k_hash", [&] {
        hipLaunchKernelGGL(ssa_k_hash, dim3(grid_for(n, 256)), dim3(256), 0, ctx->stream, ctx->d_params, d_sigs,
                           (const u8 *)ks->pks.p, mv, n, (u64 *)ctx->ws_h.p, (u8 *)nullptr, d_key_idx, (u32)ks->m);
    });
    if (rc) return rc;
    if (ks->comb)
        return timed_launch(ctx, "ssa_k_verify_keyed", [&] {
            hipLaunchKernelGGL(ssa_k_verify_keyed_comb, dim3(grid_for(n, 256)), dim3(256), 0, ctx->stream, d_sigs,
                               d_key_idx, (const u64 *)ks->ktab.p, (const u8 *)ks->status.p, (u32)ks->m,
                               (const u64 *)ctx->ws_h.p, (const u64 *)ctx->d_gtab, n, flags, d_status_out, d_fail);
        });
    return timed_launch(ctx, "ssa_k_verify_keyed", [&] {
        hipLaunchKernelGGL(ssa_k_verify_keyed, dim3(grid_for(n, 256)), dim3(256), 0, ctx->stream, d_sigs, d_key_idx,
                           (const u64 *)ks->tab.p, (const u8 *)ks->status.p, (u32)ks->m, (const u64 *)ctx->ws_h.p,
                           (const u64 *)ctx->d_gtab, n, flags, d_status_out, d_fail);
    });
}

extern "C" int ssa_verify_many_indexed(ssa_ctx *ctx, ssa_keyset *ks, const uint32_t *key_idx, const uint8_t *sigs,
                                       const uint8_t *msgs, const uint64_t *msg_off, size_t msg_stride, size_t msg_len,
                                       size_t n, uint32_t flags, uint8_t *status_out, uint64_t *n_fail_out) {
    if (!ctx || !ks || ks->ctx != ctx || (n && (!key_idx || !sigs || !status_out))) return SSA_ERR_ARG;
    if (int rc = check_msgs(msgs, msg_off, msg_stride, msg_len, n)) return rc;
    if (n_fail_out) *n_fail_out = 0;
    if (n == 0) return 0;
    HIP_TRY(hipSetDevice(ctx->device));
    StagedInputs s;
    const void *p, *p_idx;
    if (int rc = stage_up(ctx, ctx->st_sigs, sigs, n * 81, &p)) return rc;
    s.sigs = (const u8 *)p;
    if (int rc = stage_up(ctx, ctx->st_aux, key_idx, n * sizeof(uint32_t), &p_idx)) return rc;
    if (int rc = stage_msgs(ctx, msgs, msg_off, msg_stride, msg_len, n, s)) return rc;
    if (ctx->st_status.reserve(n + 16)) return SSA_ERR_HIP;
    unsigned long long *d_fail = (unsigned long long *)ctx->ws_fail.p;
    if (int rc = ssa_verify_many_indexed_device(ctx, ks, (const uint32_t *)p_idx, s.sigs, s.msgs, s.off, msg_stride,
                                                msg_len, n, flags, (u8 *)ctx->st_status.p, (uint64_t *)d_fail))
        return rc;
    HIP_TRY(hipMemcpyAsync(status_out, ctx->st_status.p, n, hipMemcpyDeviceToHost, ctx->stream));
    unsigned long long nf = 0;
    HIP_TRY(hipMemcpyAsync(&nf, d_fail, sizeof nf, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (n_fail_out) *n_fail_out = nf;
    return 0;
}

// ------------------------------------------------------------------ several GPUs from one process
// The path shards by signature with no data-path exchange (each verification reads only its own
// record), so a host caller that owns the whole batch needs no collective at all: contiguous shards,
// one context and one host thread per device, rejection counts summed on the host.
struct ssa_multi {
    std::vector<ssa_ctx *> ctxs;
};

extern "C" int ssa_multi_create(ssa_multi **out, const int *devices, int n_devices, const void *params,
                                size_t params_len) {
    if (!out || !devices || n_devices <= 0) return SSA_ERR_ARG;
    *out = nullptr;
    ssa_multi *m = new ssa_multi();
    for (int i = 0; i < n_devices; i++) {
        ssa_ctx *c = nullptr;
        int rc = ssa_ctx_create(&c, devices[i], params, params_len);
        if (rc != 0) {
            for (ssa_ctx *x : m->ctxs) ssa_ctx_destroy(x);
            delete m;
            return rc;
        }
        m->ctxs.push_back(c);
    }
    *out = m;
    return 0;
}

extern "C" void ssa_multi_destroy(ssa_multi *m) {
    if (!m) return;
    for (ssa_ctx *c : m->ctxs) ssa_ctx_destroy(c);
    delete m;
}

extern "C" int ssa_multi_verify_many(ssa_multi *m, const uint8_t *sigs, const uint8_t *pks, const uint8_t *pk_inf,
                                     const uint8_t *msgs, const uint64_t *msg_off, size_t msg_stride,
                                     size_t msg_len, size_t n, uint32_t flags, uint8_t *status_out,
                                     uint64_t *n_fail_out) {
    if (!m || m->ctxs.empty() || (n && (!sigs || !pks || !status_out))) return SSA_ERR_ARG;
    if (n_fail_out) *n_fail_out = 0;
    if (n == 0) return 0;
    const size_t world = m->ctxs.size();
    std::vector<int> rcs(world, 0);
    std::vector<uint64_t> fails(world, 0);
    std::vector<std::thread> threads;
    const size_t base = n / world, rem = n % world;
    for (size_t r = 0; r < world; r++) {
        const size_t lo = r * base + (r < rem ? r : rem), cnt = base + (r < rem ? 1 : 0);
        threads.emplace_back([&, r, lo, cnt] {
            if (cnt == 0) return;
            // shard-local message view: offsets are rebased by pointing at msgs + off[lo]
            std::vector<uint64_t> off;
            const uint8_t *mbase = msgs;
            const uint64_t *offp = nullptr;
            if (msg_off) {
                off.resize(cnt + 1);
                for (size_t k = 0; k <= cnt; k++) off[k] = msg_off[lo + k] - msg_off[lo];
                mbase = msgs + msg_off[lo];
                offp = off.data();
            } else {
                mbase = msgs ? msgs + lo * msg_stride : nullptr;
            }
            rcs[r] = ssa_verify_many(m->ctxs[r], sigs + 81 * lo, pks + 96 * lo, pk_inf ? pk_inf + lo : nullptr, mbase,
                                     offp, msg_stride, msg_len, cnt, flags, status_out + lo, &fails[r]);
        });
    }
    for (auto &t : threads) t.join();
    uint64_t total = 0;
    for (size_t r = 0; r < world; r++) {
        if (rcs[r] != 0) return rcs[r];
        total += fails[r];
    }
    if (n_fail_out) *n_fail_out = total;
    return 0;
}

// verify_batch in its MSM form over several devices: every device runs the bucket MSM on its
// contiguous shard and returns ONE point and ONE scalar; device 0 adds them up -- one Jacobian addition per shard --
// computes [sum s_i e_i]G and compares x coordinates (src/batch.rs:98-100,123-129).  The only cross-device traffic
// is 24 words per shard.
extern "C" int ssa_multi_verify_batch_msm(ssa_multi *m, const uint8_t *sigs, const uint8_t *pks, const uint8_t *pk_inf,
                                          const uint8_t *msgs, const uint64_t *msg_off, size_t msg_stride,
                                          size_t msg_len, size_t n, const uint8_t *coeffs) {
    if (!m || m->ctxs.empty() || (n && (!sigs || !pks))) return SSA_ERR_ARG;
    if (int rc = check_msgs(msgs, msg_off, msg_stride, msg_len, n)) return rc;
    if (n == 0) return SSA_OK;
    const size_t world = m->ctxs.size();
    std::vector<int> rcs(world, 0);
    std::vector<uint64_t> parts(24 * world, 0);
    std::vector<std::thread> threads;
    const size_t base = n / world, rem = n % world;
    for (size_t r = 0; r < world; r++) {
        const size_t lo = r * base + (r < rem ? r : rem), cnt = base + (r < rem ? 1 : 0);
        threads.emplace_back([&, r, lo, cnt] {
            // (an empty shard still produces its record -- the identity, 0 and the magic word: an unwritten slot is not one)
            std::vector<uint64_t> off;
            const uint8_t *mbase = msgs;
            const uint64_t *offp = nullptr;
            if (msg_off) {
                off.resize(cnt + 1);
                for (size_t k = 0; k <= cnt; k++) off[k] = msg_off[lo + k] - msg_off[lo];
                mbase = msgs + msg_off[lo];
                offp = off.data();
            } else {
                mbase = msgs ? msgs + lo * msg_stride : nullptr;
            }
            rcs[r] = ssa_verify_batch_msm_partial(m->ctxs[r], sigs + 81 * lo, pks + 96 * lo, pk_inf ? pk_inf + lo : nullptr,
                                              mbase, offp, msg_stride, msg_len, cnt, coeffs ? coeffs + 32 * lo : nullptr,
                                              &parts[24 * r]);
        });
    }
    for (auto &t : threads) t.join();
    for (size_t r = 0; r < world; r++)
        if (rcs[r] != 0) return rcs[r];
    return ssa_msm_combine(m->ctxs[0], parts.data(), world);
}

// ------------------------------------------------------------------ probes
extern "C" int ssa_debug_fault_after_chunk(ssa_ctx *ctx, int chunk) {
    if (!ctx) return SSA_ERR_ARG;
    ctx->fault_after_chunk = chunk;
    return 0;
}

extern "C" int ssa_debug_arith(ssa_ctx *ctx, int op, const uint64_t *a, const uint64_t *b, size_t n,
                               size_t a_stride, size_t b_stride, uint64_t *out, size_t out_stride) {
    if (!ctx || !a || !out || n == 0 || op < 0 || op > 18) return SSA_ERR_ARG;
    if ((op == 0 || op == 3 || op == 4 || op == 5 || (op >= 7 && op != 18)) && !b) return SSA_ERR_ARG;
    if (op == 18 && (a_stride < 2 || out_stride < 6)) return SSA_ERR_ARG;
    if (op >= 15 && op <= 17) {
        // the generated loops take their count from a[19] through v_readfirstlane (wave-uniform) and count DOWN to
        // zero: n == 0 would wrap to 2^32 iterations, rows that disagree would silently run lane 0's count
        if (a_stride < 20 || b_stride < 12 || out_stride < 19) return SSA_ERR_ARG;
        if (op != 16)
            for (size_t i = 0; i < n; i++)
                if (a[i * a_stride + 19] == 0 || a[i * a_stride + 19] > 64 || a[i * a_stride + 19] != a[19])
                    return SSA_ERR_ARG;
    }
    HIP_TRY(hipSetDevice(ctx->device));
    const void *da, *db = nullptr;
    if (int rc = stage_up(ctx, ctx->st_aux, a, n * a_stride * 8, &da)) return rc;
    if (b)
        if (int rc = stage_up(ctx, ctx->st_aux2, b, n * b_stride * 8, &db)) return rc;
    if (ctx->st_status.reserve(n * out_stride * 8)) return SSA_ERR_HIP;
    HIP_TRY(hipMemsetAsync(ctx->st_status.p, 0, n * out_stride * 8, ctx->stream));
    if (op == 7) {
        hipLaunchKernelGGL(ssa_k_debug_coop, dim3((unsigned)n), dim3(64), 0, ctx->stream, (const u64 *)da,
                           (const u64 *)db, n, a_stride, b_stride, (u64 *)ctx->st_status.p, out_stride);
    } else if (op == 4 || (op >= 15 && op <= 17)) {
        if (ctx->ws_tab.reserve(n * (size_t)(PTAB_ENTRIES * PTAB_ENTRY_U64) * sizeof(u64))) return SSA_ERR_HIP;
        hipLaunchKernelGGL(ssa_k_debug_mul, dim3(grid_for(n, 64)), dim3(64), 0, ctx->stream, op, (const u64 *)da,
                           (const u64 *)db, n, a_stride, b_stride, (u64 *)ctx->ws_tab.p,
                           (u64 *)ctx->st_status.p, out_stride);
    } else {
        hipLaunchKernelGGL(ssa_k_debug, dim3(grid_for(n, 64)), dim3(64), 0, ctx->stream, op, (const u64 *)da,
                           (const u64 *)db, n, a_stride, b_stride, (u64 *)ctx->st_status.p, out_stride);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out, ctx->st_status.p, n * out_stride * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return 0;
}

extern "C" int ssa_bench_fpmul(ssa_ctx *ctx, int variant, double *fpmul_per_s) {
    if (!ctx || !fpmul_per_s) return SSA_ERR_ARG;
    HIP_TRY(hipSetDevice(ctx->device));
    if (variant >= 10 && variant <= 13) {   // cooperative point-operation chain: result = operations per second
        if (ctx->st_aux.reserve(64)) return SSA_ERR_HIP;
        hipEvent_t c0, c1;
        HIP_TRY(hipEventCreate(&c0));
        HIP_TRY(hipEventCreate(&c1));
        const int n_ops = 2000;
        for (int rep = 0; rep < 2; rep++) {
            HIP_TRY(hipEventRecord(c0, ctx->stream));
            hipLaunchKernelGGL(ssa_k_coop_bench, dim3(1), dim3(64), 0, ctx->stream, variant - 10, n_ops,
                               (u64 *)ctx->st_aux.p);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipEventRecord(c1, ctx->stream));
            HIP_TRY(hipEventSynchronize(c1));
        }
        float cms = 0;
        HIP_TRY(hipEventElapsedTime(&cms, c0, c1));
        (void)hipEventDestroy(c0);
        (void)hipEventDestroy(c1);
        *fpmul_per_s = n_ops / (cms * 1e-3);
        return 0;
    }
    const unsigned blocks = 256 * 16, threads = 256;
    const int iters = 2000;
    if (ctx->st_aux.reserve((size_t)blocks * threads * 8)) return SSA_ERR_HIP;
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    double muls_per_thread = 0;
    for (int rep = 0; rep < 2; rep++) {  // first repetition warms up
        HIP_TRY(hipEventRecord(e0, ctx->stream));
        switch (variant) {
            case 0:
                hipLaunchKernelGGL(ssa_k_fpmul_bench<1>, dim3(blocks), dim3(threads), 0, ctx->stream,
                                   (u64 *)ctx->st_aux.p, 0x1234567ull, iters);
                muls_per_thread = 2.0 * 1 * iters;
                break;
            case 1:
                hipLaunchKernelGGL(ssa_k_fpmul_bench<4>, dim3(blocks), dim3(threads), 0, ctx->stream,
                                   (u64 *)ctx->st_aux.p, 0x1234567ull, iters);
                muls_per_thread = 2.0 * 4 * iters;
                break;
            case 2:
                hipLaunchKernelGGL(ssa_k_fpmul_bench<8>, dim3(blocks), dim3(threads), 0, ctx->stream,
                                   (u64 *)ctx->st_aux.p, 0x1234567ull, iters);
                muls_per_thread = 2.0 * 8 * iters;
                break;
            case 4:
                hipLaunchKernelGGL(ssa_k_fpsqr_bench<0>, dim3(blocks), dim3(threads), 0, ctx->stream,
                                   (u64 *)ctx->st_aux.p, 0x1234567ull, iters / 4);
                muls_per_thread = 16.0 * (iters / 4);
                break;
            case 5:
                hipLaunchKernelGGL(ssa_k_fpsqr_bench<1>, dim3(blocks), dim3(threads), 0, ctx->stream,
                                   (u64 *)ctx->st_aux.p, 0x1234567ull, iters / 4);
                muls_per_thread = 16.0 * (iters / 4);
                break;
            default:
                hipLaunchKernelGGL(ssa_k_f6mul_bench, dim3(blocks), dim3(threads), 0, ctx->stream,
                                   (u64 *)ctx->st_aux.p, 0x1234567ull, iters / 4);
                muls_per_thread = (36.0 + 21.0) * (iters / 4);
                break;
        }
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(e1, ctx->stream));
        HIP_TRY(hipEventSynchronize(e1));
    }
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *fpmul_per_s = muls_per_thread * (double)blocks * threads / (ms * 1e-3);
    return 0;
}

#ifdef SSA_WAVE_TIMES
extern "C" int ssa_debug_phase_times(unsigned long long *out, size_t n_waves) {
    if (n_waves > ssa::SSA_WAVE_TIMES_MAX) n_waves = ssa::SSA_WAVE_TIMES_MAX;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(ssa::g_phase_times), 4 * n_waves * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
// diagnostic build only: (start, end, hardware id) of every wave of the LAST ssa_k_verify launch, wall_clock64 ticks (100 MHz)
extern "C" int ssa_debug_wave_times(unsigned long long *out, size_t n_waves) {
    if (n_waves > ssa::SSA_WAVE_TIMES_MAX) n_waves = ssa::SSA_WAVE_TIMES_MAX;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(ssa::g_wave_times), 3 * n_waves * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
#endif

