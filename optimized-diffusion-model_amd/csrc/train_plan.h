// Training step on the layer plan: train-mode forward (all activations kept, dropout) and the backward pass.
// Included by rdmi.hip after the plan builder.  Kernels: bwd_kernels.h (+ the forward conv kernel for data gradients).
#pragma once

namespace {

struct BwdConv {
    int op = -1;                         // forward op index
    bool has_dgrad = false, has_sc = false, has_gn = false;
    ConvArgs dgrad{}; int dgrad_cfg = 0; // GA = dgrad(G)
    ConvArgs scgrad{}; int sc_cfg = 0;   // GS = G . Wn^T
    size_t tab_off = 0, wtab_off = 0, sctab_off = 0;
    size_t invA_start = 0, invA_list = 0, invS_start = 0, invS_list = 0;   // inverse nearest maps (int arena offsets)
    bool has_invA = false, has_invS = false;
    size_t wT_off = 0, wscT_off = 0;     // transposed packs in the backward weight arena
    int p_w = -1, p_b = -1, p_gamma = -1, p_beta = -1, p_wsc = -1, p_bsc = -1;   // parameter indices (flat-grad slices)
};

struct TrainPlan {
    bool ready = false;
    std::vector<BwdConv> convs;          // one per forward conv op (same order as c->ops; attention ops are looked up separately)
    std::vector<PackJob> jobs; std::vector<int> job_param;
    PackJob* d_jobs = nullptr;
    float* d_wb = nullptr; size_t wb_floats = 0;       // transposed weight packs
    int* d_int = nullptr;
    float* gws = nullptr;                // gradients of the activation tensors (same offsets as ws)
    // scratch of the conv backward.  G / ACT / ACTS exist 2 * group times: the weight-gradient GEMMs of a group of ops run on the
    // side stream while the main stream already produces the next group's G / ACT into the other half (two event pairs per group).
    static constexpr int MAXSETS = 16;
    int group = 8;                       // scratch sets allocated = 2 * group (RDMI_TRAIN_GROUP); conv ops per event pair at run time: group_for(NB)
    bool group_env = false;
    int group_for(int NB) const { return (group_env || NB > 256) ? group : std::min(group, 4); }   // measured at B = 128 bf16: 8 -> 3.76 ms, 4 -> 3.70, 2 -> 3.75, 1 -> 3.92
    float *G[MAXSETS] = {}, *ACT[MAXSETS] = {}, *ACTS[MAXSETS] = {};
    float *GA = nullptr, *GS = nullptr, *zero_bias = nullptr;
    hipStream_t side = nullptr; hipEvent_t ev_ready[2] = {nullptr, nullptr}, ev_done[2] = {nullptr, nullptr};
    bool two_streams = true;             // RDMI_TRAIN_STREAMS=1 keeps everything on the caller's stream
    bool fuse_colsum = true;             // RDMI_TRAIN_FUSE_COLSUM=0: every conv op launches its own bwd_scale_colsum_kernel (diagnostic)
    float *gdense = nullptr, *gta = nullptr, *gh1 = nullptr, *four = nullptr, *sig_copy = nullptr, *lab_copy = nullptr;
    std::vector<size_t> poff;            // flat-gradient offset of every parameter
    size_t ptotal = 0;
    float drop_p = 0.f; uint64_t seed = 0; int last_B = 0; int label_rows = 0;
    SgemmArgs* d_gemm_jobs = nullptr; ColsumJob* d_col_jobs = nullptr;   // job tables of the embedding backward (64 entries each)
    std::vector<SgemmArgs> h_gemm_jobs; std::vector<ColsumJob> h_col_jobs;
    std::vector<SgemmArgs> m_gemm_jobs; std::vector<ColsumJob> m_col_jobs;   // host mirror of what the device tables hold (uploads only on change)
    // ---- launch graphs (RDMI_TRAIN_GRAPH=0: plain launches).  The ~100 launches of the train-mode forward and the ~250 of the backward
    //      are each recorded ONCE (on a stream of the plan, with the weight-gradient side stream forked and joined inside) and replayed
    //      on the caller's stream by one hipGraphLaunch per step.  Everything a replay reads or writes lives at fixed addresses: the
    //      caller's x / sigma / labels / grad_out are copied into buffers of the plan first, the dropout seed is a device word.
    bool use_graph = true;
    hipStream_t cap = nullptr;                            // capture stream (the caller's may be the legacy default stream, which cannot capture)
    hipGraphExec_t fwd_exec = nullptr, bwd_exec = nullptr;
    std::vector<unsigned long long> fwd_key, bwd_key;     // what a recorded graph depends on (batch, dropout, pointers)
    int fwd_calls = 0, bwd_calls = 0;                     // the first call of each runs eagerly (one-time attribute / table setup happens there)
    float *x_in = nullptr, *out_buf = nullptr, *gout_buf = nullptr, *grads_int = nullptr;   // grads_int: the flat parameter gradient the recorded backward writes
    unsigned long long* d_seed = nullptr; unsigned long long* h_seed = nullptr; int seed_slot = 0;   // device seed word; pinned staging ring of 64
    std::vector<PackJob> m_jobs_fwd, m_jobs_bwd;          // host mirrors of the two pack-job tables (uploaded only when a parameter pointer changed)
    long graph_replays = 0, graph_records = 0;
    unsigned long long fprog_hash = 0;                    // parameter-pointer hash the training program's descriptors were last patched for
};

void conv_tile_cfg(ConvArgs& a, int& cfg) {
    if (a.Cout_pad < 32) { cfg = 3; a.S = 1; a.BN = 16; }
    else if (a.HWo >= 40) { cfg = 0; a.S = 1; a.BN = 64; }
    else if (a.HWo >= 9) { cfg = 1; a.S = std::max(1, 64 / a.HWo); a.BN = 32; }
    else { cfg = 2; a.S = std::max(1, 16 / a.HWo); a.BN = 32; }
    if (cfg == 0 && a.Cout_pad % 64 != 0) { cfg = 1; a.BN = 32; }
    a.Mpad = pad16(a.S * a.HWo);
}

}  // namespace

struct rdmi_train { TrainPlan t; };

namespace {

std::map<rdmi_ctx*, TrainPlan*>& train_registry() { static std::map<rdmi_ctx*, TrainPlan*> r; return r; }

// split the batch so that ~1000 workgroups exist whatever the layer's channel counts
int launch_wgrad(WgradArgs w, hipStream_t s) {
    if (w.lda & 3) return fail("wgrad: ACT row stride %d is not a multiple of 4", w.lda);
    if (w.HWv > 255 || w.HWo > 128) return fail("wgrad: image of %d -> %d pixels does not fit the staged chunk", w.HWv, w.HWo);
    w.S = wgrad_chunk(w.HWv, w.HWo);
    const int tiles = ceil_div(w.Cin, 32) * ceil_div(w.Cout, 64), chunks = ceil_div(w.NB, w.S);
    // how many workgroups a weight-gradient launch may occupy: it runs on the side stream BESIDE the data-gradient chain, whose short
    // kernels starve when a 512-workgroup launch holds every CU (measured: 350 us of gaps per step); fewer K splits also mean fewer
    // atomic merges of the partial tiles (HBM write traffic).  RDMI_WGRAD_WGS overrides.
    static const int wg_env = [] { const char* e = getenv("RDMI_WGRAD_WGS"); return e ? std::max(1, atoi(e)) : 0; }();
    // measured at B = 128 (bf16 step): 512 -> 3.86 ms, 256 / 128 -> 3.75 ms, 64 -> 4.16 ms; at B = 4096 the launches are long enough to want the whole chip (512: 43 ms, 128: 48 ms)
    const int wg_budget = wg_env ? wg_env : (w.NB <= 256 ? 128 : 512);
    w.ksplit = std::max(1, std::min(chunks, wg_budget / tiles));
    const dim3 grid((unsigned)w.ksplit, (unsigned)ceil_div(w.Cin, 32), (unsigned)ceil_div(w.Cout, 64));
    const size_t lds = wgrad_lds_bytes(w.HWv, w.HWo, w.S, w.bf16);
    if (w.bf16) {
        static bool attr16 = false;
        if (!attr16) {
            HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>((wgrad_mfma_kernel<9, true>)), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>((wgrad_mfma_kernel<1, true>)), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            attr16 = true;
        }
        if (w.ntap == 9) hipLaunchKernelGGL((wgrad_mfma_kernel<9, true>), grid, dim3(RDMI_THREADS), lds, s, w);
        else if (w.ntap == 1) hipLaunchKernelGGL((wgrad_mfma_kernel<1, true>), grid, dim3(RDMI_THREADS), lds, s, w);
        else return fail("wgrad: %d taps", w.ntap);
        return 0;
    }
    if (w.ntap == 9) hipLaunchKernelGGL(wgrad_mfma_kernel<9>, grid, dim3(RDMI_THREADS), lds, s, w);
    else if (w.ntap == 1) hipLaunchKernelGGL(wgrad_mfma_kernel<1>, grid, dim3(RDMI_THREADS), lds, s, w);
    else return fail("wgrad: %d taps", w.ntap);
    return 0;
}

int build_train_plan(rdmi_ctx* c, TrainPlan& T) {
    const size_t NBmax = (size_t)c->max_batch;
    // ---- every activation keeps its own storage (no liveness reuse) + a gradient twin
    {
        size_t top = 0;
        for (auto& t : c->tensors) { t.off = top; top += (t.per_sample() + 63) & ~(size_t)63; }
        c->ws_per_sample = top;
        if (c->ws) (void)hipFree(c->ws);
        HIP_OK(hipMalloc((void**)&c->ws, top * NBmax * sizeof(float)));
        HIP_OK(hipMalloc((void**)&T.gws, top * NBmax * sizeof(float)));
        for (auto& op : c->ops) {
            auto tptr = [&](int t) -> float* { return t >= 0 ? c->ws + c->tensors[(size_t)t].off * NBmax : nullptr; };
            if (op.kind == OP_CONV) {
                ConvArgs& ca = op.conv;
                ca.srcA = tptr(op.tA); ca.srcB = tptr(op.tB); ca.scA = tptr(op.tScA); ca.scB = tptr(op.tScB);
                ca.resid = tptr(op.tRes); ca.out = tptr(op.out_tensor);
            } else { op.attn.x = tptr(op.tA); op.attn.out = tptr(op.out_tensor); }
        }
        c->debug_taps = true;            // the layer plan is now the only valid plan for this context (fused plan spills differently)
    }
    // ---- compute_dtype = bf16 (BASELINE config #4): bf16 weight copies for every conv whose channel counts allow the 32-wide
    //      bf16 MFMA step (all but the 1-channel input conv); the fp32 parameters stay the master copy, accumulation is fp32
    const bool bf = c->arch.compute_dtype == 1;
    if (bf) {
        size_t w16 = 0;                                      // bf16 elements
        auto alloc16 = [&](size_t n) { size_t o = w16; w16 += (n + 127) & ~(size_t)127; return o; };
        std::vector<std::pair<size_t, size_t>> fix;          // (job index, element offset)
        for (auto& op : c->ops) {
            if (op.kind != OP_CONV) continue;
            ConvArgs& ca = op.conv;
            const ConvSpec& sp = op.spec;
            if ((ca.Cv & 31) || (ca.Csc & 31) || ca.Cv > 256) continue;   // > 256 channels: the in-place narrowing of conv_mfma_kernel needs one pass per row
            const int Cin = sp.CA + sp.CB;
            {
                PackJob j{};
                j.Cin = Cin; j.Cout = sp.Cout; j.Kpad = ca.Cv; j.Npad = ca.Cout_pad; j.n_off = 0; j.ntap = 9; j.s_co = (long)Cin * 9; j.s_ci = 9; j.s_t = 1; j.kind = 2;
                fix.push_back({c->jobs.size(), alloc16((size_t)9 * ca.Cv * ca.Cout_pad)});
                c->jobs.push_back(j); c->job_param.push_back(c->pindex.at(sp.conv + ".weight"));
            }
            if (ca.Csc) {
                const int Csc = sp.CscA + sp.CscB;
                PackJob j{};
                j.Cin = Csc; j.Cout = sp.Cout; j.Kpad = ca.Csc; j.Npad = ca.Cout_pad; j.n_off = 0; j.ntap = 1; j.s_co = 1; j.s_ci = sp.Cout; j.s_t = 0; j.kind = 2;
                fix.push_back({c->jobs.size(), alloc16((size_t)ca.Csc * ca.Cout_pad)});
                c->jobs.push_back(j); c->job_param.push_back(c->pindex.at(sp.nin + ".W"));
            }
            ca.bf16 = 1;
        }
        // activation storage: the workspace tensors are bf16 (same element offsets, half the bytes); the caller's x and out stay fp32
        for (auto& op : c->ops) {
            if (op.kind == OP_CONV) { op.conv.a_bf16 = op.a_is_input ? 0 : 1; op.conv.b_bf16 = 1; op.conv.o_bf16 = op.out_is_output ? 0 : 1; }
            else op.attn.io_bf16 = 1;
        }
        HIP_OK(hipMalloc((void**)&c->d_w16, std::max<size_t>(w16, 64) * sizeof(bf16_t)));
        HIP_OK(hipMemset(c->d_w16, 0, std::max<size_t>(w16, 64) * sizeof(bf16_t)));
        size_t k = 0;
        for (auto& op : c->ops) {
            if (op.kind != OP_CONV || !op.conv.bf16) continue;
            c->jobs[fix[k].first].dst = reinterpret_cast<float*>(c->d_w16 + fix[k].second);
            op.conv.wpk = reinterpret_cast<const float*>(c->d_w16 + fix[k].second); ++k;
            if (op.conv.Csc) {
                c->jobs[fix[k].first].dst = reinterpret_cast<float*>(c->d_w16 + fix[k].second);
                op.conv.wsc = reinterpret_cast<const float*>(c->d_w16 + fix[k].second); ++k;
            }
        }
        if (c->d_jobs) (void)hipFree(c->d_jobs);
        HIP_OK(hipMalloc((void**)&c->d_jobs, c->jobs.size() * sizeof(PackJob)));
        c->packed_valid = false;
    }
    T.poff.resize(c->params.size());
    T.ptotal = 0;
    for (size_t i = 0; i < c->params.size(); ++i) { T.poff[i] = T.ptotal; T.ptotal += c->params[i].numel; }

    std::vector<int> ints;
    size_t wb = 0;
    auto alloc_wb = [&](size_t n) { size_t o = wb; wb += (n + 63) & ~(size_t)63; return o; };
    size_t maxG = 1, maxV = 1, maxS = 1;
    auto gptr = [&](int t) -> float* { return t >= 0 ? T.gws + c->tensors[(size_t)t].off * NBmax : nullptr; };
    (void)gptr;
    T.convs.clear();
    for (size_t oi = 0; oi < c->ops.size(); ++oi) {
        const Op& op = c->ops[oi];
        if (op.kind != OP_CONV) continue;
        const ConvSpec& sp = op.spec;
        const ConvArgs& fa = op.conv;
        BwdConv b;
        b.op = (int)oi;
        const int Cin = sp.CA + sp.CB;
        b.has_gn = !sp.gn.empty();
        b.has_dgrad = !op.a_is_input;
        b.has_sc = fa.Csc > 0;
        b.p_w = c->pindex.at(sp.conv + ".weight"); b.p_b = c->pindex.at(sp.conv + ".bias");
        if (b.has_gn) { b.p_gamma = c->pindex.at(sp.gn + ".weight"); b.p_beta = c->pindex.at(sp.gn + ".bias"); }
        if (b.has_sc) { b.p_wsc = c->pindex.at(sp.nin + ".W"); b.p_bsc = c->pindex.at(sp.nin + ".b"); }
        maxG = std::max(maxG, (size_t)fa.HWo * pad16(sp.Cout));
        maxV = std::max(maxV, (size_t)fa.HWv * fa.Cv);
        if (b.has_sc) maxS = std::max(maxS, (size_t)fa.HWo * fa.Csc);
        // forward tap table per sample: [HWo][9] -> virtual input pixel or -1 (weight gradient)
        b.wtab_off = ints.size();
        for (int o = 0; o < fa.HWo; ++o) {
            const int oy = o / sp.Wo, ox = o % sp.Wo;
            for (int t = 0; t < 9; ++t) {
                const int iy = oy * sp.stride + t / 3 - sp.pad_lo, ix = ox * sp.stride + t % 3 - sp.pad_lo;
                ints.push_back((iy >= 0 && iy < sp.Hv && ix >= 0 && ix < sp.Wv) ? iy * sp.Wv + ix : -1);
            }
        }
        if (b.has_dgrad) {
            // data gradient = conv over G (grid Ho x Wo, Cout channels) producing the virtual-input grid (Hv x Wv, Cin channels)
            ConvArgs& d = b.dgrad;
            d = ConvArgs{};
            d.CA = sp.Cout; d.CB = 0; d.Cv = pad16(sp.Cout);
            d.HWa = fa.HWo; d.HWv = fa.HWo; d.HWo = fa.HWv; d.ntap = 9; d.G = 0;
            d.Cout = Cin; d.Cout_pad = pad16(Cin); d.out_scale = 1.f; d.eps = 1e-6f;
            conv_tile_cfg(d, b.dgrad_cfg);
            if (conv_lds_bytes(d) > 160 * 1024) return fail("backward of %s: LDS tile too large", sp.name.c_str());
            // adjoint table: row m = s*HWv + v (virtual input pixel), tap t -> LDS row s*HWo + o with in(o, t) == v
            b.tab_off = ints.size();
            const int zrow = d.S * d.HWv;                // (d.HWv is the dgrad conv's input = forward output grid)
            for (int m = 0; m < d.Mpad; ++m) {
                const int sidx = m / d.HWo, v = m % d.HWo;
                const bool real = m < d.S * d.HWo;
                const int vy = v / sp.Wv, vx = v % sp.Wv;
                for (int t = 0; t < 9; ++t) {
                    int row = zrow;
                    const int ny = vy - (t / 3 - sp.pad_lo), nx = vx - (t % 3 - sp.pad_lo);
                    if (real && ny >= 0 && nx >= 0 && ny % sp.stride == 0 && nx % sp.stride == 0) {
                        const int oy = ny / sp.stride, ox = nx / sp.stride;
                        if (oy < sp.Ho && ox < sp.Wo) row = sidx * d.HWv + oy * sp.Wo + ox;
                    }
                    ints.push_back(row);
                }
                ints.push_back(zrow);                    // shortcut column unused
            }
            // transposed weights: K = co, N = ci : packed[t][co/16][ci_pad][16] = W[co][ci][t]
            b.wT_off = alloc_wb((size_t)9 * d.Cv * d.Cout_pad);
            PackJob j{};
            j.dst = reinterpret_cast<float*>(b.wT_off);
            j.Cin = sp.Cout; j.Cout = Cin; j.Kpad = d.Cv; j.Npad = d.Cout_pad; j.n_off = 0; j.ntap = 9;
            j.s_co = 9; j.s_ci = (long)Cin * 9; j.s_t = 1; j.kind = 0;     // "co" of the job = ci of W, "ci" of the job = co of W
            if (bf && (d.Cv & 31) == 0 && d.Cv <= 256) { j.kind = 2; d.bf16 = 1; }
            if (bf) { d.a_bf16 = 1; d.o_bf16 = 1; }
            T.jobs.push_back(j); T.job_param.push_back(b.p_w);
        }
        if (b.has_sc) {
            ConvArgs& d = b.scgrad;
            d = ConvArgs{};
            const int Csc = sp.CscA + sp.CscB;
            d.CA = sp.Cout; d.CB = 0; d.Cv = pad16(sp.Cout);
            d.HWa = fa.HWo; d.HWv = fa.HWo; d.HWo = fa.HWo; d.ntap = 1; d.G = 0;
            d.Cout = Csc; d.Cout_pad = pad16(Csc); d.out_scale = 1.f; d.eps = 1e-6f;
            conv_tile_cfg(d, b.sc_cfg);
            b.sctab_off = ints.size();
            for (int m = 0; m < d.Mpad; ++m) { ints.push_back(m < d.S * d.HWo ? m : d.S * d.HWv); ints.push_back(d.S * d.HWv); }
            b.wscT_off = alloc_wb((size_t)d.Cv * d.Cout_pad);
            PackJob j{};
            j.dst = reinterpret_cast<float*>(b.wscT_off);
            j.Cin = sp.Cout; j.Cout = Csc; j.Kpad = d.Cv; j.Npad = d.Cout_pad; j.n_off = 0; j.ntap = 1;
            j.s_co = sp.Cout; j.s_ci = 1; j.s_t = 0; j.kind = 0;            // NIN W [in=Csc][out=Cout]: job "co" = in, job "ci" = out
            if (bf && (d.Cv & 31) == 0) { j.kind = 2; d.bf16 = 1; }
            if (bf) { d.a_bf16 = 1; d.o_bf16 = 1; }
            T.jobs.push_back(j); T.job_param.push_back(b.p_wsc);
        }
        // inverse nearest maps for the scatter of gradients back to mapped sources
        auto inverse = [&](const std::vector<int>& map, int HWsrc, size_t& st, size_t& li) {
            std::vector<std::vector<int>> inv((size_t)HWsrc);
            for (size_t v = 0; v < map.size(); ++v) inv[(size_t)map[v]].push_back((int)v);
            st = ints.size();
            int acc = 0;
            for (int s2 = 0; s2 < HWsrc; ++s2) { ints.push_back(acc); acc += (int)inv[(size_t)s2].size(); }
            ints.push_back(acc);
            li = ints.size();
            for (auto& l : inv) for (int v : l) ints.push_back(v);
        };
        if (op.has_mapA) { inverse(op.mapA, fa.HWa, b.invA_start, b.invA_list); b.has_invA = true; }
        if (op.has_mapSc) { inverse(op.mapSc, fa.HWsa, b.invS_start, b.invS_list); b.has_invS = true; }
        T.convs.push_back(b);
    }
    T.wb_floats = std::max<size_t>(wb, 64);
    HIP_OK(hipMalloc((void**)&T.d_wb, T.wb_floats * sizeof(float)));
    HIP_OK(hipMemset(T.d_wb, 0, T.wb_floats * sizeof(float)));
    HIP_OK(hipMalloc((void**)&T.d_int, std::max<size_t>(ints.size(), 1) * sizeof(int)));
    HIP_OK(hipMemcpy(T.d_int, ints.data(), ints.size() * sizeof(int), hipMemcpyHostToDevice));
    HIP_OK(hipMalloc((void**)&T.d_jobs, std::max<size_t>(T.jobs.size(), 1) * sizeof(PackJob)));
    for (auto& j : T.jobs) j.dst = T.d_wb + reinterpret_cast<size_t>(j.dst);
    if (const char* e = getenv("RDMI_TRAIN_STREAMS")) T.two_streams = atoi(e) != 1;
    if (const char* e = getenv("RDMI_TRAIN_FUSE_COLSUM")) T.fuse_colsum = atoi(e) != 0;
    if (const char* e = getenv("RDMI_TRAIN_GROUP")) { T.group = std::max(1, std::min(TrainPlan::MAXSETS / 2, atoi(e))); T.group_env = true; }
    if (!T.two_streams) T.group = 1;
    for (int p = 0; p < 2 * T.group; ++p) {
        HIP_OK(hipMalloc((void**)&T.G[p], maxG * NBmax * sizeof(float)));
        HIP_OK(hipMalloc((void**)&T.ACT[p], maxV * NBmax * sizeof(float)));
        HIP_OK(hipMalloc((void**)&T.ACTS[p], maxS * NBmax * sizeof(float)));
    }
    for (int p = 0; p < 2; ++p) {
        HIP_OK(hipEventCreateWithFlags(&T.ev_ready[p], hipEventDisableTiming));
        HIP_OK(hipEventCreateWithFlags(&T.ev_done[p], hipEventDisableTiming));
    }
    if (T.two_streams) {             // RDMI_TRAIN_SIDE_PRIO=1: lowest stream priority for the weight gradients.  Off by default: measured 8.0 ms per step
        int lo = 0, hi = 0;          // instead of 3.9 (the low-priority queue is starved outright on this runtime, it does not merely yield)
        const char* pe = getenv("RDMI_TRAIN_SIDE_PRIO");
        if (pe && atoi(pe) != 0 && hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess && lo != hi) { if (hipStreamCreateWithPriority(&T.side, hipStreamNonBlocking, lo) != hipSuccess) T.side = nullptr; }
        if (!T.side) HIP_OK(hipStreamCreateWithFlags(&T.side, hipStreamNonBlocking));
    }
    HIP_OK(hipMalloc((void**)&T.GA, maxV * NBmax * sizeof(float)));
    HIP_OK(hipMalloc((void**)&T.GS, maxS * NBmax * sizeof(float)));
    HIP_OK(hipMalloc((void**)&T.zero_bias, 1024 * sizeof(float)));
    HIP_OK(hipMemset(T.zero_bias, 0, 1024 * sizeof(float)));
    const size_t Mp = (size_t)pad16(c->max_batch);
    HIP_OK(hipMalloc((void**)&T.gdense, Mp * c->dense_total * sizeof(float)));
    HIP_OK(hipMalloc((void**)&T.gta, Mp * c->temb * sizeof(float)));
    HIP_OK(hipMalloc((void**)&T.gh1, Mp * c->temb * sizeof(float)));
    HIP_OK(hipMalloc((void**)&T.four, Mp * 2 * c->arch.nf * sizeof(float)));
    HIP_OK(hipMalloc((void**)&T.d_gemm_jobs, 64 * sizeof(SgemmArgs)));
    HIP_OK(hipMalloc((void**)&T.d_col_jobs, 64 * sizeof(ColsumJob)));
    T.h_gemm_jobs.reserve(64); T.h_col_jobs.reserve(64);
    HIP_OK(hipMalloc((void**)&T.sig_copy, Mp * sizeof(float)));
    HIP_OK(hipMalloc((void**)&T.lab_copy, Mp * std::max(1, c->arch.num_classes) * sizeof(float)));
    {
        const size_t E = (size_t)c->H * c->W * c->arch.channels;
        HIP_OK(hipMalloc((void**)&T.x_in, Mp * E * sizeof(float)));
        HIP_OK(hipMalloc((void**)&T.out_buf, Mp * E * sizeof(float)));
        HIP_OK(hipMalloc((void**)&T.gout_buf, Mp * E * sizeof(float)));
        HIP_OK(hipMalloc((void**)&T.grads_int, std::max<size_t>(T.ptotal, 1) * sizeof(float)));
        HIP_OK(hipMalloc((void**)&T.d_seed, 64));
        HIP_OK(hipMemset(T.d_seed, 0, 64));
        HIP_OK(hipHostMalloc((void**)&T.h_seed, 64 * sizeof(unsigned long long), 0));
        if (const char* e = getenv("RDMI_TRAIN_GRAPH")) T.use_graph = atoi(e) != 0;
        if (T.use_graph && hipStreamCreateWithFlags(&T.cap, hipStreamNonBlocking) != hipSuccess) { T.cap = nullptr; T.use_graph = false; }
    }
    // resolve pointers of the data-gradient convs
    for (auto& b : T.convs) {
        if (b.has_dgrad) {
            ConvArgs& d = b.dgrad;
            d.srcA = nullptr; d.tab = T.d_int + b.tab_off; d.wpk = T.d_wb + b.wT_off; d.bias = T.zero_bias; d.out = T.GA;   // srcA = G[parity] at launch
        }
        if (b.has_sc) {
            ConvArgs& d = b.scgrad;
            d.srcA = nullptr; d.tab = T.d_int + b.sctab_off; d.wpk = T.d_wb + b.wscT_off; d.bias = T.zero_bias; d.out = T.GS;
        }
    }
    T.ready = true;
    return 0;
}

TrainPlan* get_train(rdmi_ctx* c) {
    auto& r = train_registry();
    auto it = r.find(c);
    return it == r.end() ? nullptr : it->second;
}

}  // namespace

extern "C" {

int rdmi_enable_training(rdmi_ctx* c) {
    if (!c) return fail("null context");
    if (get_train(c)) return 0;
    TrainPlan* T = new TrainPlan();
    int e = 0;
    try { e = build_train_plan(c, *T); } catch (const std::exception& ex) { e = fail("training plan: %s", ex.what()); }
    if (e) { delete T; return e; }
    train_registry()[c] = T;
    // The training forward as ONE workgroup-resident launch (the S = 1 fused program + a stash of every layer output + Dropout_0
    // in the GroupNorm_1 epilogues) instead of ~100 layer-plan launches.  RDMI_TRAIN_FUSED=0, or a shape the planner cannot fit,
    // keeps the layer plan's forward.
    const char* tf = getenv("RDMI_TRAIN_FUSED");
    if (c->fused_ready() && (!tf || atoi(tf) != 0)) {
        if (int e2 = build_one_program(c, 1, false, true)) return e2;
        if (c->progs.back().ok && c->progs.back().train) c->train_prog = (int)c->progs.size() - 1;
        T->fprog_hash = 0;
    }
    return 0;
}

}  // extern "C"

namespace {

unsigned long long param_ptr_hash(const rdmi_ctx* c);

// Pack-job tables follow the parameter pointers (torch keeps the storage): uploaded only when one changed; the parameter-dependent
// pointers of the layer plan are refreshed on the host.  No launch, no synchronisation.
int train_refresh_params(rdmi_ctx* c, TrainPlan& T, hipStream_t s) {
    for (auto& p : c->params)
        if (!p.ptr) return fail("parameter '%s' was never bound (rdmi_set_param)", p.name.c_str());
    for (size_t i = 0; i < c->jobs.size(); ++i) {
        c->jobs[i].src = c->params[(size_t)c->job_param[i]].ptr;
        const int p2 = i < c->job_param2.size() ? c->job_param2[i] : -1;
        c->jobs[i].src2 = p2 >= 0 ? c->params[(size_t)p2].ptr : nullptr;
    }
    for (size_t i = 0; i < T.jobs.size(); ++i) T.jobs[i].src = c->params[(size_t)T.job_param[i]].ptr;
    auto same = [](const std::vector<PackJob>& a, const std::vector<PackJob>& b) { return a.size() == b.size() && (a.empty() || std::memcmp(a.data(), b.data(), a.size() * sizeof(PackJob)) == 0); };
    if (!same(T.m_jobs_fwd, c->jobs)) {
        T.m_jobs_fwd = c->jobs;
        HIP_OK(hipMemcpyAsync(c->d_jobs, T.m_jobs_fwd.data(), T.m_jobs_fwd.size() * sizeof(PackJob), hipMemcpyHostToDevice, s));
    }
    if (!T.jobs.empty() && !same(T.m_jobs_bwd, T.jobs)) {
        T.m_jobs_bwd = T.jobs;
        HIP_OK(hipMemcpyAsync(T.d_jobs, T.m_jobs_bwd.data(), T.m_jobs_bwd.size() * sizeof(PackJob), hipMemcpyHostToDevice, s));
    }
    for (auto& op : c->ops) {
        if (op.kind == OP_CONV) {
            ConvArgs& a = op.conv;
            a.bias = P(c, op.p_bias);
            a.bias_sc = op.p_bias_sc.empty() ? nullptr : P(c, op.p_bias_sc);
            a.gamma = op.p_gamma.empty() ? nullptr : P(c, op.p_gamma);
            a.beta = op.p_beta.empty() ? nullptr : P(c, op.p_beta);
        } else { op.attn.gamma = P(c, op.p_gamma); op.attn.beta = P(c, op.p_beta); op.attn.b3 = P(c, op.p_b3); }
    }
    if (c->train_prog >= 0) {
        rdmi_ctx::FusedProg& q = c->progs[(size_t)c->train_prog];
        const unsigned long long h = param_ptr_hash(c) | 1ull;
        if (q.ok && T.fprog_hash != h) {
            for (auto& f : q.fpatch) {
                FOp& o = q.fprog[(size_t)f.op];
                const float* p = f.param.empty() ? c->d_w + f.arena_off : P(c, f.param);
                switch (f.field) {
                    case FusedBuilder::F_GAMMA: o.gamma = p; break;
                    case FusedBuilder::F_BETA: o.beta = p; break;
                    case FusedBuilder::F_BIAS: case FusedBuilder::F_BIAS_ARENA: o.bias = p; break;
                    case FusedBuilder::F_BIAS2: o.bias2 = p; break;
                    case FusedBuilder::F_W: o.main_ph.w = p; break;
                    case FusedBuilder::F_SC0W: o.sc[0].w = p; break;
                    case FusedBuilder::F_SC1W: o.sc[1].w = p; break;
                }
            }
            HIP_OK(hipMemcpyAsync(q.d_fprog, q.fprog.data(), q.fprog.size() * sizeof(FOp), hipMemcpyHostToDevice, s));
            T.fprog_hash = h;
        }
    }
    c->packed_valid = true;
    return 0;
}

unsigned long long param_ptr_hash(const rdmi_ctx* c) {
    unsigned long long h = 1469598103934665603ull;
    for (auto& p : c->params) { h ^= (unsigned long long)reinterpret_cast<size_t>(p.ptr); h *= 1099511628211ull; }
    return h;
}

// Run `body(stream)`: replay its recorded graph when `key` is unchanged, record it when it is not (from the second call on), or
// launch it eagerly (first call, RDMI_TRAIN_GRAPH=0, or a runtime that cannot capture).
template <class Body>
int run_recorded(TrainPlan& T, hipGraphExec_t& exec, std::vector<unsigned long long>& have, const std::vector<unsigned long long>& key, int& calls,
                 hipStream_t s, Body&& body) {
    if (T.use_graph && exec && have == key) { ++T.graph_replays; HIP_OK(hipGraphLaunch(exec, s)); return 0; }
    if (!T.use_graph || calls++ < 1) return body(s);
    if (exec) { (void)hipGraphExecDestroy(exec); exec = nullptr; have.clear(); }
    // everything already enqueued on the caller's stream (input copies, seed) must precede the graph: the graph is launched on `s`
    if (hipStreamBeginCapture(T.cap, hipStreamCaptureModeThreadLocal) != hipSuccess) { (void)hipGetLastError(); T.use_graph = false; return body(s); }
    const int e = body(T.cap);
    hipGraph_t g = nullptr;
    const hipError_t ce = hipStreamEndCapture(T.cap, &g);
    if (e) { if (g) (void)hipGraphDestroy(g); return e; }
    if (ce != hipSuccess || !g) { (void)hipGetLastError(); T.use_graph = false; return body(s); }
    const hipError_t ie = hipGraphInstantiate(&exec, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    if (ie != hipSuccess) { (void)hipGetLastError(); exec = nullptr; T.use_graph = false; return body(s); }
    have = key; ++T.graph_records;
    HIP_OK(hipGraphLaunch(exec, s));
    return 0;
}

}  // namespace

extern "C" {

// Train-mode forward: layer plan, every activation kept, Dropout_0 (p) on the input of every Conv_1.
int rdmi_train_forward(rdmi_ctx* c, const float* x, const float* sigma, const float* labels, float* out, int B, float dropout_p,
                       uint64_t seed, void* stream) {
    if (!c || !x || !sigma || !out) return fail("null argument");
    TrainPlan* T = get_train(c);
    if (!T) return fail("call rdmi_enable_training first");
    if (B < 1 || B > c->max_batch) return fail("batch %d outside [1, %d]", B, c->max_batch);
    hipStream_t s = (hipStream_t)stream;
    if (int e = train_refresh_params(c, *T, s)) return e;
    T->drop_p = dropout_p; T->seed = seed; T->last_B = B;
    // inputs -> fixed addresses; the seed -> the device word the kernels read
    const size_t E = (size_t)c->H * c->W * c->arch.channels;
    HIP_OK(hipMemcpyAsync(T->x_in, x, (size_t)B * E * sizeof(float), hipMemcpyDeviceToDevice, s));
    HIP_OK(hipMemcpyAsync(T->sig_copy, sigma, (size_t)B * sizeof(float), hipMemcpyDeviceToDevice, s));
    if (labels) HIP_OK(hipMemcpyAsync(T->lab_copy, labels, (size_t)B * c->arch.num_classes * sizeof(float), hipMemcpyDeviceToDevice, s));
    T->seed_slot = (T->seed_slot + 1) & 63;
    T->h_seed[T->seed_slot] = seed;
    HIP_OK(hipMemcpyAsync(T->d_seed, T->h_seed + T->seed_slot, sizeof(unsigned long long), hipMemcpyHostToDevice, s));
    auto body = [&](hipStream_t ss) -> int {
        for (size_t oi = 0; oi < c->ops.size(); ++oi) {
            Op& op = c->ops[oi];
            if (op.kind == OP_CONV) { op.conv.drop_p = op.dropout ? dropout_p : 0.f; op.conv.drop_seed = seed; op.conv.seed_dev = T->d_seed; op.conv.op_id = (uint32_t)oi; }
        }
        hipLaunchKernelGGL(pack_kernel, dim3(32, (unsigned)c->jobs.size()), dim3(RDMI_THREADS), 0, ss, (const PackJob*)c->d_jobs);
        FwdIn f{T->x_in, 0, T->sig_copy, 0, 0.f, 0, 0.f, 0.f, labels ? T->lab_copy : nullptr, B, T->out_buf, B};
        const bool keep = c->use_fused;
        c->use_fused = false;                          // (the layer plan, unless the training program exists: run_forward looks at train_prog)
        c->in_train_forward = true;
        c->train_drop_p = dropout_p; c->train_seed_dev = T->d_seed;
        const int e = run_forward(c, f, ss);
        c->in_train_forward = false;
        c->use_fused = keep;
        for (auto& op : c->ops) if (op.kind == OP_CONV) { op.conv.drop_p = 0.f; op.conv.seed_dev = nullptr; }
        return e;
    };
    unsigned pb; std::memcpy(&pb, &dropout_p, 4);
    const std::vector<unsigned long long> key{(unsigned long long)B, pb, labels ? 1ull : 0ull, param_ptr_hash(c), c->profiling ? 1ull : 0ull};
    if (c->profiling) { if (int e = body(s)) return e; }         // per-launch events: never from a recorded graph
    else if (int e = run_recorded(*T, T->fwd_exec, T->fwd_key, key, T->fwd_calls, s, body)) return e;
    HIP_OK(hipMemcpyAsync(out, T->out_buf, (size_t)B * E * sizeof(float), hipMemcpyDeviceToDevice, s));
    return 0;
}

// Backward of the last rdmi_train_forward: grad_out [B,1,H,W] -> every parameter gradient, written (not accumulated) into
// grads_flat in the reference's parameter order (offsets = running sum of numel; time_embed.W stays zero).
int rdmi_backward(rdmi_ctx* c, const float* grad_out, float* grads_flat, size_t grads_numel, const float* x, void* stream) {
    if (!c || !grad_out || !grads_flat || !x) return fail("null argument");
    TrainPlan* Tp = get_train(c);
    if (!Tp) return fail("call rdmi_enable_training first");
    TrainPlan& T = *Tp;
    if (grads_numel != T.ptotal) return fail("grads buffer holds %zu floats, the model has %zu parameters", grads_numel, T.ptotal);
    hipStream_t s0 = (hipStream_t)stream;
    const int NB = T.last_B;
    const size_t NBmax = (size_t)c->max_batch;
    const int sbf = c->arch.compute_dtype == 1;      // bf16 activation workspace and scratch tensors
    if (int e = train_refresh_params(c, T, s0)) return e;
    x = T.x_in;                                      // the forward's own copy of the input (fixed address)
    {
        const size_t E = (size_t)c->H * c->W * c->arch.channels;
        HIP_OK(hipMemcpyAsync(T.gout_buf, grad_out, (size_t)NB * E * sizeof(float), hipMemcpyDeviceToDevice, s0));
        grad_out = T.gout_buf;
    }
    float* const grads_caller = grads_flat;
    grads_flat = T.grads_int;                        // fixed address: the recorded launches write here, one copy hands the result to the caller
    auto body = [&](hipStream_t s) -> int {
    // transposed packs
    if (!T.jobs.empty()) hipLaunchKernelGGL(pack_kernel, dim3(32, (unsigned)T.jobs.size()), dim3(RDMI_THREADS), 0, s, (const PackJob*)T.d_jobs);
    HIP_OK(hipMemsetAsync(grads_flat, 0, T.ptotal * sizeof(float), s));
    HIP_OK(hipMemsetAsync(T.gws, 0, c->ws_per_sample * NBmax * sizeof(float), s));
    HIP_OK(hipMemsetAsync(T.gdense, 0, (size_t)pad16(c->max_batch) * c->dense_total * sizeof(float), s));
    auto gptr = [&](int t) -> float* { return t >= 0 ? T.gws + c->tensors[(size_t)t].off * NBmax : nullptr; };
    auto pgrad = [&](int pi) -> float* { return pi >= 0 ? grads_flat + T.poff[(size_t)pi] : nullptr; };
    static bool attr = false;
    if (!attr) {
        HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(gn_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_mfma_kernel<9>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_mfma_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_kernel<64>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr = true;
    }
    // map forward op index -> backward conv record
    std::map<int, const BwdConv*> bmap;
    for (auto& b : T.convs) bmap[b.op] = &b;

    hipStream_t s2 = T.two_streams ? T.side : s;           // weight-gradient stream
    const int grp = T.group_for(NB);
    bool done_rec[2] = {false, false};
    int nconv = 0;                                          // conv ops seen: set = nconv % (2 * group), group half = set / group
    std::vector<WgradArgs> pending;                         // weight gradients of the current group
    auto flush_wgrads = [&](int half) -> int {
        if (pending.empty()) return 0;
        if (T.two_streams) { HIP_OK(hipEventRecord(T.ev_ready[half], s)); HIP_OK(hipStreamWaitEvent(s2, T.ev_ready[half], 0)); }
        for (auto& w : pending) if (int e = launch_wgrad(w, s2)) return e;
        if (T.two_streams) { HIP_OK(hipEventRecord(T.ev_done[half], s2)); done_rec[half] = true; }
        pending.clear();
        return 0;
    };
    bool colsum_fused = false;                              // the current op's bwd_scale_colsum ran as the tail of the previous op's gn_bwd
    auto colsum_args = [&](int oj, int pj) {
        const Op& o2 = c->ops[(size_t)oj];
        const BwdConv& b2 = *bmap.at(oj);
        ColsumArgs ca{};
        ca.gY = o2.out_is_output ? grad_out : gptr(o2.out_tensor); ca.G = T.G[pj]; ca.gR = gptr(o2.tRes); ca.scale = o2.conv.out_scale;
        ca.gdense = o2.use_dense ? T.gdense : (float*)nullptr; ca.dense_stride = c->dense_total; ca.dense_off = o2.conv.dense_off;
        ca.db = pgrad(b2.p_b); ca.db2 = pgrad(b2.p_bsc); ca.HW = o2.conv.HWo; ca.C = o2.spec.Cout; ca.g_bf16 = sbf;
        return ca;
    };
    for (int oi = (int)c->ops.size() - 1; oi >= 0; --oi) {
        const Op& op = c->ops[(size_t)oi];
        if (op.kind == OP_ATTN) {
            AttnBwdArgs a{};
            a.x = op.attn.x; a.gOut = gptr(op.out_tensor); a.gX = gptr(op.tA);
            a.gamma = op.attn.gamma; a.beta = op.attn.beta;
            a.dgamma = pgrad(c->pindex.at(op.name + ".GroupNorm_0.weight")); a.dbeta = pgrad(c->pindex.at(op.name + ".GroupNorm_0.bias"));
            for (int k = 0; k < 4; ++k) {
                const int pw = c->pindex.at(op.name + ".NIN_" + std::to_string(k) + ".W"), pb = c->pindex.at(op.name + ".NIN_" + std::to_string(k) + ".b");
                a.W[k] = c->params[(size_t)pw].ptr; a.b[k] = c->params[(size_t)pb].ptr; a.dW[k] = pgrad(pw); a.db[k] = pgrad(pb);
            }
            a.NB = NB; a.L = op.attn.L; a.G = op.attn.G; a.eps = op.attn.eps; a.scale = op.attn.scale; a.out_scale = op.attn.out_scale; a.x_bf16 = sbf;
            hipLaunchKernelGGL(attn_bwd_kernel<64>, dim3((unsigned)std::min(NB, 256)), dim3(AB_THREADS), attn_bwd_lds_bytes<64>(a.L, a.G), s, a);
            HIP_OK(hipGetLastError());
            continue;
        }
        const BwdConv& b = *bmap.at(oi);
        const ConvSpec& sp = op.spec;
        const ConvArgs& fa = op.conv;
        const int Cin = sp.CA + sp.CB;
        const float* gY = op.out_is_output ? grad_out : gptr(op.out_tensor);
        const int p = nconv % (2 * grp), half = p / grp;
        ++nconv;
        float* Gp = T.G[p]; float* ACTp = T.ACT[p]; float* ACTSp = T.ACTS[p];
        if (T.two_streams && p % grp == 0 && done_rec[half] && !colsum_fused) HIP_OK(hipStreamWaitEvent(s, T.ev_done[half], 0));   // the group before last has left this half
        // G = scale * gY (+ identity residual) and the bias / NIN-bias / Dense_0 gradients (column sums of G): its own launch only where the
        // previous backward kernel was not a GroupNorm backward that could carry it as its tail (first op, ops behind an attention block)
        if (!colsum_fused) {
            const ColsumArgs ca = colsum_args(oi, p);
            hipLaunchKernelGGL(bwd_scale_colsum_kernel, dim3((unsigned)NB), dim3(RDMI_THREADS), 0, s, ca);
        }
        colsum_fused = false;
        // the next conv op's first kernel rides on this op's last GroupNorm backward
        ColsumArgs tail{};
        if (T.fuse_colsum && oi > 0 && c->ops[(size_t)oi - 1].kind != OP_ATTN) {
            const int pn = nconv % (2 * grp), hn = pn / grp;           // (nconv already counts this op)
            if (T.two_streams && pn % grp == 0 && done_rec[hn]) HIP_OK(hipStreamWaitEvent(s, T.ev_done[hn], 0));   // its buffer set must be free before the tail writes G
            tail = colsum_args(oi - 1, pn);
            colsum_fused = true;
        }
        // data gradient w.r.t. the activated input
        if (b.has_dgrad) {
            ConvArgs d = b.dgrad; d.NB = NB; d.srcA = Gp;
            if (int e = launch_conv(b.dgrad_cfg, d, s)) return e;
        }
        // GroupNorm / SiLU / dropout backward (+ materialise ACT for the weight gradient)
        {
            GnBwdArgs g{};
            g.srcA = op.a_is_input ? x : fa.srcA; g.srcB = fa.srcB; g.mapA = fa.mapA;
            g.CA = fa.CA; g.CB = fa.CB; g.Cv = fa.Cv; g.HWa = fa.HWa; g.HWv = fa.HWv; g.srcA_mod = 0; g.NB = NB;
            g.GA = T.GA; g.ACT = ACTp; g.a_bf16 = op.a_is_input ? 0 : sbf; g.b_bf16 = sbf; g.s_bf16 = sbf;
            g.has_gn = b.has_gn ? 1 : 0; g.G = fa.G; g.eps = fa.eps;
            if (b.has_gn) { g.gamma = fa.gamma; g.beta = fa.beta; g.dgamma = pgrad(b.p_gamma); g.dbeta = pgrad(b.p_beta); }
            g.drop_p = op.dropout ? T.drop_p : 0.f; g.seed = T.seed; g.seed_dev = T.d_seed; g.op_id = (uint32_t)oi;
            if (b.has_dgrad) {
                g.gA = gptr(op.tA); g.gB = gptr(op.tB);
                if (b.has_invA) { g.inv_start = T.d_int + b.invA_start; g.inv_list = T.d_int + b.invA_list; }
            }
            const size_t lds = gn_bwd_lds_bytes(fa.HWv, fa.Cv);
            if (lds > 160 * 1024) return fail("gn backward of %s: LDS %zu B", sp.name.c_str(), lds);
            if (!b.has_sc) g.tail = tail;
            hipLaunchKernelGGL(gn_bwd_kernel, dim3((unsigned)NB), dim3(GN_THREADS), lds, s, g);
        }
        // NIN shortcut: data gradient and its scatter (ACTS = the shortcut's input, for its weight gradient)
        if (b.has_sc) {
            ConvArgs d = b.scgrad; d.NB = NB; d.srcA = Gp;
            if (int e = launch_conv(b.sc_cfg, d, s)) return e;
            GnBwdArgs g{};
            g.srcA = fa.scA; g.srcB = fa.scB; g.mapA = fa.mapSc;
            g.CA = fa.CscA; g.CB = fa.CscB; g.Cv = fa.Csc; g.HWa = fa.HWsa; g.HWv = fa.HWo; g.NB = NB;
            g.GA = T.GS; g.ACT = ACTSp; g.has_gn = 0; g.a_bf16 = sbf; g.b_bf16 = sbf; g.s_bf16 = sbf;
            g.gA = gptr(op.tScA); g.gB = gptr(op.tScB);
            if (b.has_invS) { g.inv_start = T.d_int + b.invS_start; g.inv_list = T.d_int + b.invS_list; }
            g.tail = tail;
            hipLaunchKernelGGL(gn_bwd_kernel, dim3((unsigned)NB), dim3(GN_THREADS), gn_bwd_lds_bytes(fa.HWo, fa.Csc), s, g);
        }
        // weight gradients on the side stream (reference OIHW layout): dW[co][ci][t] += sum ACT[in(o,t)][ci] G[o][co];  NIN: dWn += Vs^T G
        {
            WgradArgs w{};
            w.ACT = ACTp; w.G = Gp; w.dW = pgrad(b.p_w); w.tab = T.d_int + b.wtab_off;
            w.NB = NB; w.HWv = fa.HWv; w.HWo = fa.HWo; w.Cin = Cin; w.Cout = sp.Cout; w.ntap = 9;
            w.lda = fa.Cv;                                   // ACT has Cv (padded) channels per pixel; only ci < Cin are real
            w.s_co = (long)Cin * 9; w.s_ci = 9; w.s_t = 1;
            w.bf16 = sbf; w.s_bf16 = sbf;
            pending.push_back(w);
        }
        if (b.has_sc) {
            WgradArgs w{};
            const int Csc = sp.CscA + sp.CscB;
            w.ACT = ACTSp; w.G = Gp; w.dW = pgrad(b.p_wsc); w.tab = nullptr;
            w.NB = NB; w.HWv = fa.HWo; w.HWo = fa.HWo; w.Cin = Csc; w.Cout = sp.Cout; w.ntap = 1; w.lda = fa.Csc;
            w.s_co = 1; w.s_ci = sp.Cout; w.s_t = 0;                    // NIN W [in][out]
            w.bf16 = sbf; w.s_bf16 = sbf;
            pending.push_back(w);
        }
        if (p % grp == grp - 1) { if (int e = flush_wgrads(half)) return e; }
        HIP_OK(hipGetLastError());
    }
    if (int e = flush_wgrads(((nconv - 1) % (2 * grp)) / grp)) return e;

    // ---- embedding backward: Dense_0 (x17) -> SiLU -> [label_emb, time_mlp.2] -> SiLU -> time_mlp.0.  Independent GEMMs and
    //      bias column sums of one stage go out as one job-table launch each (tables staged in T.h_*: they live until the next call).
    const int Tm = c->temb, DT = c->dense_total, nf = c->arch.nf;
    {
        auto& GJ = T.h_gemm_jobs; auto& CJ = T.h_col_jobs;
        GJ.clear(); CJ.clear();
        size_t g_used = 0, c_used = 0;                         // entries of the device tables already consumed by earlier launches
        T.m_gemm_jobs.resize(64); T.m_col_jobs.resize(64);
        auto flush_gemm = [&]() -> int {
            const size_t nj = GJ.size() - g_used;
            if (!nj) return 0;
            if (GJ.size() > 64) return fail("embedding backward: %zu GEMM jobs", GJ.size());
            int mm = 0, mn = 0, mk = 0;
            for (size_t i = g_used; i < GJ.size(); ++i) { mm = std::max(mm, GJ[i].M); mn = std::max(mn, GJ[i].N); if (!GJ[i].no_split) mk = std::max(mk, GJ[i].K); }
            const int ks = std::max(1, std::min(ceil_div(std::max(mk, 1), 64), 16));
            if (std::memcmp(T.m_gemm_jobs.data() + g_used, GJ.data() + g_used, nj * sizeof(SgemmArgs)) != 0) {   // same buffers as last step: already resident
                std::memcpy(T.m_gemm_jobs.data() + g_used, GJ.data() + g_used, nj * sizeof(SgemmArgs));
                HIP_OK(hipMemcpyAsync(T.d_gemm_jobs + g_used, GJ.data() + g_used, nj * sizeof(SgemmArgs), hipMemcpyHostToDevice, s));
            }
            hipLaunchKernelGGL(small_gemm_jobs_kernel, dim3((unsigned)ceil_div(mm, 64), (unsigned)ceil_div(mn, 64), (unsigned)(nj * ks)), dim3(RDMI_THREADS), 0, s,
                               (const SgemmArgs*)(T.d_gemm_jobs + g_used), ks);
            g_used = GJ.size();
            return 0;
        };
        auto flush_col = [&](int M, int ldx) -> int {
            const size_t nj = CJ.size() - c_used;
            if (!nj) return 0;
            if (CJ.size() > 64) return fail("embedding backward: %zu column-sum jobs", CJ.size());
            int mc = 0;
            for (size_t i = c_used; i < CJ.size(); ++i) mc = std::max(mc, CJ[i].C);
            if (std::memcmp(T.m_col_jobs.data() + c_used, CJ.data() + c_used, nj * sizeof(ColsumJob)) != 0) {
                std::memcpy(T.m_col_jobs.data() + c_used, CJ.data() + c_used, nj * sizeof(ColsumJob));
                HIP_OK(hipMemcpyAsync(T.d_col_jobs + c_used, CJ.data() + c_used, nj * sizeof(ColsumJob), hipMemcpyHostToDevice, s));
            }
            hipLaunchKernelGGL(colsum_jobs_kernel, dim3((unsigned)ceil_div(mc, 64), (unsigned)std::max(1, std::min(64, M / 64)), (unsigned)nj), dim3(RDMI_THREADS), 0, s,
                               (const ColsumJob*)(T.d_col_jobs + c_used), M, ldx);
            c_used = CJ.size();
            return 0;
        };
        Layout L = build_layout(c);
        std::vector<std::pair<std::string, int>> blocks;
        for (auto& d : L.down) blocks.push_back({d.name, d.cout});
        blocks.push_back({"mid_block1", L.mid_ch}); blocks.push_back({"mid_block2", L.mid_ch});
        for (auto& u : L.up) blocks.push_back({u.name, u.cout});
        HIP_OK(hipMemsetAsync(T.gta, 0, (size_t)pad16(c->max_batch) * Tm * sizeof(float), s));
        int off = 0;
        for (auto& bl : blocks) {
            const int pw = c->pindex.at(bl.first + ".Dense_0.weight"), pb = c->pindex.at(bl.first + ".Dense_0.bias");
            SgemmArgs g{};   // dWd[co][k] = sum_n gdense[n][off+co] * silu(temb[n][k])
            g.A = T.gdense + off; g.a_m = 1; g.a_k = DT; g.a_act = 0;
            g.B = c->d_temb; g.b_k = Tm; g.b_n = 1; g.b_act = 1;
            g.C = pgrad(pw); g.c_m = Tm; g.c_n = 1; g.accumulate = 0; g.M = bl.second; g.N = Tm; g.K = NB;
            GJ.push_back(g);
            CJ.push_back(ColsumJob{T.gdense + off, pgrad(pb), bl.second, 0});
            SgemmArgs h{};   // gta[n][k] += sum_co gdense[n][off+co] * Wd[co][k]   (gta zeroed above; the 17 blocks add with atomics)
            h.A = T.gdense + off; h.a_m = DT; h.a_k = 1; h.B = c->params[(size_t)pw].ptr; h.b_k = Tm; h.b_n = 1;
            h.C = T.gta; h.c_m = Tm; h.c_n = 1; h.accumulate = 1; h.M = NB; h.N = Tm; h.K = bl.second; h.no_split = 1;
            GJ.push_back(h);
            off += bl.second;
        }
        if (int e = flush_gemm()) return e;
        if (int e = flush_col(NB, DT)) return e;
        hipLaunchKernelGGL(silu_bwd_kernel, dim3((unsigned)ceil_div(NB * Tm, RDMI_THREADS)), dim3(RDMI_THREADS), 0, s, T.gta, (const float*)c->d_temb, (long)NB * Tm);
        // gta is now g(temb)
        if (c->arch.conditional) {
            const int pw = c->pindex.at("label_emb.weight"), pb = c->pindex.at("label_emb.bias"), nc = c->arch.num_classes;
            SgemmArgs g{};   // dWl[k][cl] = sum_n gtemb[n][k] * labels[n][cl]
            g.A = T.gta; g.a_m = 1; g.a_k = Tm; g.B = T.lab_copy; g.b_k = nc; g.b_n = 1; g.C = pgrad(pw); g.c_m = nc; g.c_n = 1;
            g.M = Tm; g.N = nc; g.K = NB;
            GJ.push_back(g);
            CJ.push_back(ColsumJob{T.gta, pgrad(pb), Tm, 0});
        }
        {
            const int pw = c->pindex.at("time_mlp.2.weight"), pb = c->pindex.at("time_mlp.2.bias");
            SgemmArgs g{};   // dW2[k][j] = sum_n gtemb[n][k] * silu(h1[n][j])
            g.A = T.gta; g.a_m = 1; g.a_k = Tm; g.B = c->d_h1; g.b_k = Tm; g.b_n = 1; g.b_act = 1; g.C = pgrad(pw); g.c_m = Tm; g.c_n = 1;
            g.M = Tm; g.N = Tm; g.K = NB;
            GJ.push_back(g);
            CJ.push_back(ColsumJob{T.gta, pgrad(pb), Tm, 0});
            SgemmArgs h{};   // gh1[n][j] = sum_k gtemb[n][k] * W2[k][j]
            h.A = T.gta; h.a_m = Tm; h.a_k = 1; h.B = c->params[(size_t)pw].ptr; h.b_k = Tm; h.b_n = 1; h.C = T.gh1; h.c_m = Tm; h.c_n = 1;
            h.M = NB; h.N = Tm; h.K = Tm; h.no_split = 1;
            GJ.push_back(h);
            if (int e = flush_gemm()) return e;
            if (int e = flush_col(NB, Tm)) return e;
            hipLaunchKernelGGL(silu_bwd_kernel, dim3((unsigned)ceil_div(NB * Tm, RDMI_THREADS)), dim3(RDMI_THREADS), 0, s, T.gh1, (const float*)c->d_h1, (long)NB * Tm);
        }
        {
            const int pw = c->pindex.at("time_mlp.0.weight"), pb = c->pindex.at("time_mlp.0.bias");
            hipLaunchKernelGGL(fourier_kernel, dim3((unsigned)ceil_div(NB * 2 * nf, RDMI_THREADS)), dim3(RDMI_THREADS), 0, s, (const float*)T.sig_copy,
                               c->params[(size_t)c->pindex.at("time_embed.W")].ptr, T.four, NB, nf);
            SgemmArgs g{};   // dW0[j][f] = sum_n gh1[n][j] * four[n][f]
            g.A = T.gh1; g.a_m = 1; g.a_k = Tm; g.B = T.four; g.b_k = 2 * nf; g.b_n = 1; g.C = pgrad(pw); g.c_m = 2 * nf; g.c_n = 1;
            g.M = Tm; g.N = 2 * nf; g.K = NB;
            GJ.push_back(g);
            CJ.push_back(ColsumJob{T.gh1, pgrad(pb), Tm, 0});
            if (int e = flush_gemm()) return e;
            if (int e = flush_col(NB, Tm)) return e;
        }
        HIP_OK(hipGetLastError());
    }
    if (T.two_streams)                                      // join: the caller's stream continues after the last weight gradients
        for (int p = 0; p < 2; ++p) if (done_rec[p]) HIP_OK(hipStreamWaitEvent(s, T.ev_done[p], 0));
    return 0;
    };   // body
    unsigned pb; std::memcpy(&pb, &T.drop_p, 4);
    const std::vector<unsigned long long> key{(unsigned long long)NB, pb, param_ptr_hash(c)};
    if (int e = run_recorded(T, T.bwd_exec, T.bwd_key, key, T.bwd_calls, s0, body)) return e;
    HIP_OK(hipMemcpyAsync(grads_caller, T.grads_int, T.ptotal * sizeof(float), hipMemcpyDeviceToDevice, s0));
    return 0;
}

// Diagnostic: how often the training step's launch graphs were recorded and replayed (tests: the recorded path is what runs).
int rdmi_train_graph_stats(rdmi_ctx* c, long* records, long* replays) {
    TrainPlan* T = c ? get_train(c) : nullptr;
    if (!T || !records || !replays) return fail("no training plan");
    *records = T->graph_records; *replays = T->graph_replays;
    return 0;
}

}  // extern "C"
