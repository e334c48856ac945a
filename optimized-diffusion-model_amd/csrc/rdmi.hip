// librdmi: host side of the MI355X-native NCSN++ / reflected PC sampler (C ABI in include/rdmi.h).
//
// Structure: rdmi_create() turns the architecture keys into a static LAUNCH PLAN -- a list of fused
// kernels over a liveness-packed NHWC activation workspace -- mirroring the data flow of
// NCSNpp.forward (RD/models/ncsnpp.py:226-354).  Entry points replay the plan on the caller's stream.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <climits>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/rdmi.h"
#include "attn_kernel.h"
#include "conv_kernel.h"
#include "misc_kernels.h"
#include "unet_kernel.h"
#include "bwd_kernels.h"
#include "opt_kernels.h"
#include "ode_kernels.h"
#include "tiled_kernels.h"

namespace {

thread_local std::string g_err;

int fail(const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return 1;
}

#define HIP_OK(expr)                                                                             \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess) return fail("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
inline int pad16(int a) { return (a + 15) & ~15; }

struct Param {
    std::string name;
    std::vector<int> shape;
    size_t numel = 0;
    const float* ptr = nullptr;
};

struct Tensor {
    std::string name;
    int C = 0, H = 0, W = 0;
    size_t off = 0;           // floats per sample from the workspace base (times max_batch)
    int def = -1, last = -1;  // defining / last-reading op index
    size_t per_sample() const { return (size_t)C * H * W; }
};

// description of one fused conv launch of the layer plan (kept per op: the backward plan is derived from it)
struct ConvSpec {
    std::string name;
    int tA = -1, tB = -1;           // sources (tA == -2: caller's x)
    int CA = 0, CB = 0;
    int Ha = 0, Wa = 0;             // source-A dims
    int Hv = 0, Wv = 0;             // virtual input dims
    std::string gn;                 // GroupNorm param prefix ("" = none)
    std::string conv;               // conv param prefix
    int stride = 1, pad_lo = 1;
    int Ho = 0, Wo = 0, Cout = 0;
    // shortcut
    int tScA = -1, tScB = -1, CscA = 0, CscB = 0, Hsa = 0, Wsa = 0;
    std::string nin;
    int dense_off = -1;
    int tRes = -1;
    float scale = 1.f;
    bool to_output = false;
    bool dropout = false;
};

enum OpKind { OP_CONV, OP_ATTN };

struct Op {
    OpKind kind;
    std::string name;
    ConvArgs conv{};
    AttnArgs attn{};
    int cfg = 0;                 // conv tile configuration
    int out_tensor = -1;
    double flops_per_sample = 0;
    // tensor ids, resolved to pointers once the workspace is allocated
    int tA = -1, tB = -1, tScA = -1, tScB = -1, tRes = -1;
    bool a_is_input = false;     // A source is the caller's x (input_conv)
    bool out_is_output = false;  // writes the caller's out (out_conv)
    bool use_dense = false;      // epilogue adds this block's Dense_0(SiLU(temb)) column slice
    std::vector<int> tab, mapA, mapSc;
    size_t tab_off = 0, mapA_off = 0, mapSc_off = 0;   // into the int arena
    bool has_mapA = false, has_mapSc = false;
    // packed weight locations (floats into the weight arena)
    size_t w_off = 0, wsc_off = 0, w3_off = 0, bqkv_off = 0;
    std::string p_gamma, p_beta, p_bias, p_bias_sc, p_b3;
    ConvSpec spec;               // OP_CONV: what was asked for
    int attn_C = 0, attn_H = 0, attn_W = 0;
    bool dropout = false;        // train mode applies Dropout_0 to this op's activated input (Conv_1 of a resblock)
};

struct ProfEntry { std::string name; double ms = 0; long launches = 0; double flops = 0; };

}  // namespace

struct rdmi_ctx {
    rdmi_arch arch{};
    int max_batch = 0, H = 0, W = 0;
    int temb = 0, dense_total = 0;
    std::vector<Param> params;
    std::map<std::string, int> pindex;
    std::vector<Tensor> tensors;
    std::vector<Op> ops;
    std::vector<PackJob> jobs;          // host copy (src pointers patched from params before upload)
    std::vector<int> job_param;         // param index feeding each job
    std::vector<int> job_param2;        // second source of a job (PackJob::src2), -1: none; filled up to jobs.size() in do_repack
    PackJob* d_jobs = nullptr;
    float* d_w = nullptr;               // packed weight arena
    size_t w_floats = 0;
    int* d_int = nullptr;               // tables and pixel maps
    float* ws = nullptr;                // activation workspace
    size_t ws_per_sample = 0;
    // embedding path
    size_t w_t0 = 0, w_t2 = 0, w_dense = 0, b_dense = 0;
    float *d_h1 = nullptr, *d_temb = nullptr, *d_dense = nullptr;
    // sampler scratch
    float *d_s2 = nullptr, *d_score = nullptr, *d_z = nullptr, *d_norms = nullptr, *d_ts = nullptr, *d_tvec = nullptr;
    float *d_tt = nullptr, *d_th1 = nullptr; int tt_cap = 0;   // sampler: time-path rows of all updates (rdmi_pc_sample)
    float* d_dense_all = nullptr; size_t dense_all_cap = 0;    // sampler: Dense_0 outputs of a chunk of updates [U][NBm][dense_total]
    int ts_cap = 0;
    StepState* d_state = nullptr;
    // fused (workgroup-resident) path
    bool fused_ok = false, use_fused = true;
    std::string fused_why;               // why the fused program could not be built (falls back to the layer plan)
    std::vector<FOp> fprog;              // host copy; parameter pointers are patched at repack time
    struct FPatch { int op; int field; std::string param; size_t arena_off; };
    std::vector<FPatch> fpatch;
    std::vector<short> ftabs;
    FOp* d_fprog = nullptr; short* d_ftabs = nullptr;
    float* d_spill = nullptr; size_t spill_per_sample = 0;
    UnetArgs fargs{};
    long long* d_stamps = nullptr;       // diagnostic (RDMI_STAMPS=1)
    std::vector<std::string> fdesc;      // one line per fused op
    size_t fused_lds = 0;
    // (the fields above are the CONSTRUCTION state of one program; finished programs live here, one per samples-per-workgroup)
    struct FusedProg {
        int S = 1; bool ok = false; std::string why;
        std::vector<FOp> fprog; std::vector<FPatch> fpatch; std::vector<short> ftabs; std::vector<std::string> fdesc;
        FOp* d_fprog = nullptr; short* d_ftabs = nullptr; float* d_spill = nullptr; size_t spill_per_sample = 0;
        UnetArgs fargs{}; size_t fused_lds = 0;
        bool coop = false; int n_xchg = 0; unsigned long long* d_xbuf = nullptr; int* d_coop_err = nullptr; int max_nb = 0;     // co-operative program
        bool train = false;                  // the training forward's program (stashes every layer output, applies Dropout_0): never picked for inference
    };
    // tiled plan (shapes whose samples do not fit one workgroup: csrc/tiled_kernels.h)
    struct TLaunch {
        int kind = 0;                 // 0 conv, 1 GroupNorm statistics (pass over the tensor), 2 batched GEMM, 3 softmax, 4 transpose, 5 statistics from channel sums,
                                      // 6 GroupNorm(+SiLU) written once as bf16 for a pre-activated conv (bf16 plan), 7 fused softmax(QK^T)V (bf16 plan)
        std::string name;
        TConvArgs conv{}; int nmt = 4; bool pre = false;          // pre: input is the bf16 tensor a kind-6 launch wrote (tconv_pre_kernel)
        size_t oOut2 = (size_t)-1;                                                                              // kind 6: second output (raw bf16 copy)
        GnActArgs gact{}; bool fin = false; size_t oCsA = (size_t)-1, oCsB = (size_t)-1;                        // kind 6 (fin: statistics from the producers' channel records inside the launch)
        FlashArgs flash{}; int flashC = 0;                                                                      // kind 7: fused attention core (bf16 plan)
        const float *sA = nullptr, *sB = nullptr; int CA = 0, CB = 0, HW = 0, G = 0; float* stats = nullptr;     // kind 1
        BgemmArgs gemm{};                                                                                       // kind 2
        float* sm = nullptr; long rows_per_sample = 0; int L = 0;                                               // kind 3
        const float* tsrc = nullptr; float* tdst = nullptr; int tL = 0, tC = 0, tld = 0, tc0 = 0; bool t16 = false;   // kind 4 (t16: bf16 elements)
        int pxA = 0, pxB = 0;                                                                                   // kind 5: pixels per full tile of each producer
        // per-sample workspace offsets (floats) of the operands, resolved to pointers by finish_tiled_plan: NONE = absent, XIN = the NHWC input copy
        static constexpr size_t NONE = (size_t)-1, XIN = (size_t)-2;
        size_t oA = NONE, oB = NONE, oStats = NONE, oResid = NONE, oOut = NONE, oC = NONE; long dA = 0, dB = 0;   // dA/dB: interior deltas of GEMM operands
        bool use_dense = false;
        std::string p_gamma, p_beta, p_bias; size_t w_off = 0; size_t bias_arena = (size_t)-1;
        bool in_is_x = false, out_is_final = false;
        double flops_per_sample = 0;
    };
    bool tiled = false;
    bool in_train_forward = false;       // set around run_forward by rdmi_train_forward
    bf16_t* d_w16 = nullptr;             // bf16 copies of the forward conv weights (training with compute_dtype = bf16)
    std::vector<TLaunch> tl;
    float *t_ws = nullptr, *t_xin = nullptr, *t_out = nullptr; size_t t_ws_per_sample = 0;
    long tiled_min_wgs = 512;                           // a tiled conv widens its workgroups (NCT column tiles per wave) while the launch keeps this many (RDMI_TILED_MIN_WGS: tests)
    long iconv_min_wgs = LONG_MAX;                      // iconv_kernel from this many 128 x 128 tiles per launch (off unless RDMI_ICONV=1 / RDMI_ICONV_MIN_WGS)
    void* t_zero = nullptr;                             // 256 zero bytes: iconv_kernel's source for window pixels outside the image
    std::vector<FusedProg> progs;
    int s_min_wg = 256;                            // a program with S samples per workgroup is used from batch s_min_wg * S (RDMI_S_MIN_WG: tests)
    bool use_coop = true;                          // RDMI_COOP=0: never select the co-operative program (A/B, tests)
    int coop_stride = 1;                           // ids of a group's members are this far apart (RDMI_COOP_STRIDE).  1: under round-robin placement member m of every
                                                   // group sits on XCDs m and m + 4, so an XCD's L2 only ever holds ITS quarter of the shared weights (2.5 MB: it stays
                                                   // resident from one launch to the next); 8 would put a whole group on one XCD (cheaper exchanges, 4x the weights per L2).
                                                   // Measured at B = 128: 821 vs 837 us per update.  Speed only: the exchange is correct under any placement.
    unsigned coop_epoch = 0;                       // tag base of the next co-operative launch
    int last_prog = 0;                             // index in progs of the program the last forward ran (diagnostics)
    const FusedProg* pick(int NB) const {
        // a batch that leaves CUs to spare at one sample per workgroup and fits the chip in ONE wave of workgroups: the co-operative
        // program (groups of four CUs share the low-resolution weights); larger batches: the program with the most samples per
        // workgroup that still fills the chip
        if (use_coop) for (auto& q : progs) if (q.ok && q.coop && !q.train && NB <= q.max_nb) return &q;
        const FusedProg* best = nullptr;
        for (auto& q : progs) if (q.ok && !q.coop && !q.train && (q.S == 1 || NB >= s_min_wg * q.S) && (!best || q.S > best->S)) best = &q;
        return best;
    }
    bool fused_ready() const { return !progs.empty() && progs[0].ok; }
    float train_drop_p = 0.f; const unsigned long long* train_seed_dev = nullptr;      // set around the training forward's launch
    int train_prog = -1;                           // index in progs of the training forward's program (-1: the layer plan runs the training forward)
    int cur_train = 0;
    int cur_coop = 0, cur_nxchg = 0, cur_cap_n = 0; unsigned long long* cur_xbuf = nullptr; int* cur_coop_err = nullptr;      // construction state (see stash_program)
    std::map<std::string, size_t> wmap;  // packed-weight arena offsets by parameter prefix
    bool packed_valid = false;
    bool debug_taps = false;
    bool profiling = false;
    std::vector<ProfEntry> prof;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool;
    std::vector<int> ev_entry;
    size_t ev_used = 0;
};

namespace {

// ------------------------------------------------------------------------------------------
// parameter list: the reference's state-dict order (RD/models/ncsnpp.py:42-224)
// ------------------------------------------------------------------------------------------
void add_param(rdmi_ctx* c, const std::string& name, std::vector<int> shape) {
    Param p;
    p.name = name;
    p.shape = shape;
    p.numel = 1;
    for (int s : shape) p.numel *= (size_t)s;
    c->pindex[name] = (int)c->params.size();
    c->params.push_back(p);
}
void add_gn(rdmi_ctx* c, const std::string& pre, int ch) { add_param(c, pre + ".weight", {ch}); add_param(c, pre + ".bias", {ch}); }
void add_conv_p(rdmi_ctx* c, const std::string& pre, int cin, int cout) { add_param(c, pre + ".weight", {cout, cin, 3, 3}); add_param(c, pre + ".bias", {cout}); }
void add_nin(rdmi_ctx* c, const std::string& pre, int cin, int cout) { add_param(c, pre + ".W", {cin, cout}); add_param(c, pre + ".b", {cout}); }
void add_resblock_p(rdmi_ctx* c, const std::string& pre, int cin, int cout) {
    add_gn(c, pre + ".GroupNorm_0", cin);
    add_conv_p(c, pre + ".Conv_0", cin, cout);
    add_param(c, pre + ".Dense_0.weight", {cout, c->temb});
    add_param(c, pre + ".Dense_0.bias", {cout});
    add_gn(c, pre + ".GroupNorm_1", cout);
    add_conv_p(c, pre + ".Conv_1", cout, cout);
    if (cin != cout) add_nin(c, pre + ".NIN_0", cin, cout);
}
void add_attn_p(rdmi_ctx* c, const std::string& pre, int ch) {
    add_gn(c, pre + ".GroupNorm_0", ch);
    for (int i = 0; i < 4; ++i) add_nin(c, pre + ".NIN_" + std::to_string(i), ch, ch);
}

struct BlockSpec { std::string name; int cin, cout; bool attn; int level; };

struct Layout {
    std::vector<BlockSpec> down, up;       // in execution order
    std::vector<int> skip_ch;
    int mid_ch = 0;
};

Layout build_layout(rdmi_ctx* c) {
    const rdmi_arch& a = c->arch;
    Layout L;
    int in_ch = a.nf, d = 0;
    for (int i = 0; i < a.n_levels; ++i) {
        const int out_ch = a.nf * a.ch_mult[i];
        for (int j = 0; j < a.num_res_blocks; ++j) {
            L.down.push_back({"down_blocks." + std::to_string(d), in_ch, out_ch, (a.attn_levels >> i & 1) != 0, i});
            in_ch = out_ch;
            L.skip_ch.push_back(in_ch);
            ++d;
        }
        L.skip_ch.push_back(in_ch);
    }
    L.mid_ch = in_ch;
    std::vector<int> sk(L.skip_ch.rbegin(), L.skip_ch.rend());
    int u = 0, k = 0;
    for (int i = a.n_levels - 1; i >= 0; --i) {
        const int out_ch = a.nf * a.ch_mult[i];
        for (int j = 0; j < a.num_res_blocks + 1; ++j) {
            L.up.push_back({"up_blocks." + std::to_string(u), in_ch + sk[k++], out_ch, (a.attn_levels >> i & 1) != 0, i});
            in_ch = out_ch;
            ++u;
        }
    }
    return L;
}

void build_params(rdmi_ctx* c, const Layout& L) {
    const rdmi_arch& a = c->arch;
    add_param(c, "time_embed.W", {a.nf});
    add_param(c, "time_mlp.0.weight", {c->temb, 2 * a.nf});
    add_param(c, "time_mlp.0.bias", {c->temb});
    add_param(c, "time_mlp.2.weight", {c->temb, c->temb});
    add_param(c, "time_mlp.2.bias", {c->temb});
    if (a.conditional) {
        add_param(c, "label_emb.weight", {c->temb, a.num_classes});
        add_param(c, "label_emb.bias", {c->temb});
    }
    add_conv_p(c, "input_conv", a.channels, a.nf);
    for (auto& b : L.down) add_resblock_p(c, b.name, b.cin, b.cout);
    for (size_t i = 0; i < L.down.size(); ++i)
        if (L.down[i].attn) add_attn_p(c, "down_attn." + std::to_string(i), L.down[i].cout);
    {
        int ch = a.nf;
        for (int i = 0; i < a.n_levels; ++i) {
            ch = a.nf * a.ch_mult[i];
            if (i != a.n_levels - 1) add_conv_p(c, "downsample." + std::to_string(i) + ".Conv_0", ch, ch);
        }
    }
    add_resblock_p(c, "mid_block1", L.mid_ch, L.mid_ch);
    add_resblock_p(c, "mid_block2", L.mid_ch, L.mid_ch);
    for (auto& b : L.up) add_resblock_p(c, b.name, b.cin, b.cout);
    for (size_t i = 0; i < L.up.size(); ++i)
        if (L.up[i].attn) add_attn_p(c, "up_attn." + std::to_string(i), L.up[i].cout);
    for (int k = 0, i = a.n_levels - 1; i >= 1; --i, ++k) {
        const int ch = a.nf * a.ch_mult[i];
        add_conv_p(c, "upsample." + std::to_string(k) + ".Conv_0", ch, ch);
    }
    add_gn(c, "out_norm", a.nf * a.ch_mult[0]);
    add_conv_p(c, "out_conv", a.nf * a.ch_mult[0], a.channels);
}

// ------------------------------------------------------------------------------------------
// plan builder
// ------------------------------------------------------------------------------------------
struct Builder {
    rdmi_ctx* c;
    std::vector<int> ints;       // table / map arena (host)
    size_t wf = 0;               // packed weight floats so far

    int new_tensor(const std::string& name, int C, int H, int W) {
        Tensor t;
        t.name = name; t.C = C; t.H = H; t.W = W;
        t.def = (int)c->ops.size();
        c->tensors.push_back(t);
        return (int)c->tensors.size() - 1;
    }
    void use(int t) { if (t >= 0) c->tensors[t].last = (int)c->ops.size(); }

    size_t alloc_w(size_t n) { size_t o = wf; wf += (n + 63) & ~(size_t)63; return o; }

    void job_pack(const std::string& pname, size_t dst_off, int Cin, int Cout, int Kpad, int Npad, int n_off, int ntap,
                  long s_co, long s_ci, long s_t) {
        PackJob j{};
        j.dst = reinterpret_cast<float*>(dst_off);   // patched to a pointer after the arena exists
        j.Cin = Cin; j.Cout = Cout; j.Kpad = Kpad; j.Npad = Npad; j.n_off = n_off; j.ntap = ntap;
        j.s_co = s_co; j.s_ci = s_ci; j.s_t = s_t; j.kind = 0;
        c->jobs.push_back(j);
        c->job_param.push_back(c->pindex.at(pname));
    }
    void job_copy(const std::string& pname, size_t dst_off, int n, int n_off) {
        PackJob j{};
        j.dst = reinterpret_cast<float*>(dst_off);
        j.Cout = n; j.n_off = n_off; j.kind = 1;
        c->jobs.push_back(j);
        c->job_param.push_back(c->pindex.at(pname));
    }
    // conv weight OIHW -> [9][Kpad/16][Npad][16]
    size_t pack_conv(const std::string& pre, int cin, int cout) {
        const int Kp = pad16(cin), Np = pad16(cout);
        size_t o = alloc_w((size_t)9 * Kp * Np);
        job_pack(pre + ".weight", o, cin, cout, Kp, Np, 0, 9, (long)cin * 9, 9, 1);
        c->wmap[pre] = o;
        return o;
    }
    // NIN W [in][out] -> [Kpad/16][Npad][16]
    size_t pack_nin(const std::string& pre, int cin, int cout) {
        const int Kp = pad16(cin), Np = pad16(cout);
        size_t o = alloc_w((size_t)Kp * Np);
        job_pack(pre + ".W", o, cin, cout, Kp, Np, 0, 1, 1, cout, 0);
        c->wmap[pre] = o;
        return o;
    }

    static std::vector<int> nearest_map(int Hs, int Ws, int Hd, int Wd) {
        // F.interpolate(mode='nearest'): src = floor(dst * in / out)   (RD/models/ncsnpp.py:320, layerspp.py:122)
        std::vector<int> m((size_t)Hd * Wd);
        for (int y = 0; y < Hd; ++y)
            for (int x = 0; x < Wd; ++x) {
                int sy = std::min((int)std::floor(y * ((float)Hs / Hd)), Hs - 1);
                int sx = std::min((int)std::floor(x * ((float)Ws / Wd)), Ws - 1);
                m[(size_t)y * Wd + x] = sy * Ws + sx;
            }
        return m;
    }

    int add_conv(const ConvSpec& s, int* err) {
        Op op;
        op.kind = OP_CONV;
        op.name = s.name;
        ConvArgs& a = op.conv;
        const int Cin = s.CA + s.CB;
        a.CA = s.CA; a.CB = s.CB; a.Cv = pad16(Cin);
        a.HWa = s.Ha * s.Wa; a.HWv = s.Hv * s.Wv; a.HWo = s.Ho * s.Wo;
        a.ntap = 9;
        a.G = s.gn.empty() ? 0 : std::min(Cin / 4, 32);
        a.Cout = s.Cout; a.Cout_pad = pad16(s.Cout);
        a.out_scale = s.scale; a.eps = 1e-6f;
        a.dense_stride = c->dense_total; a.dense_off = s.dense_off < 0 ? 0 : s.dense_off;
        a.CscA = s.CscA; a.CscB = s.CscB; a.Csc = pad16(s.CscA + s.CscB) * ((s.CscA + s.CscB) > 0);
        a.HWsa = s.Hsa * s.Wsa;
        op.tA = s.tA; op.tB = s.tB; op.tScA = s.tScA; op.tScB = s.tScB; op.tRes = s.tRes;
        op.a_is_input = (s.tA == -2);
        op.out_is_output = s.to_output;
        op.use_dense = s.dense_off >= 0;
        op.spec = s;
        op.dropout = s.dropout;
        // tile configuration
        if (a.Cout_pad < 32) { op.cfg = 3; a.S = 1; a.BN = 16; }                   // <4,1,1,2,1>
        else if (a.HWo >= 40) { op.cfg = 0; a.S = 1; a.BN = 64; }                  // <1,4,1,6,1>
        else if (a.HWo >= 9) { op.cfg = 1; a.S = std::max(1, 64 / a.HWo); a.BN = 32; }   // <2,2,1,3,1>
        else { op.cfg = 2; a.S = std::max(1, 16 / a.HWo); a.BN = 32; }             // <1,2,2,1,1>
        a.Mpad = pad16(a.S * a.HWo);
        const int cap[4] = {96, 96, 16, 128};
        if (a.Mpad > cap[op.cfg] || a.Cout_pad % a.BN != 0 || (op.cfg == 2 && (a.Cv >> 4) < 2)) { *err = fail("conv %s: unsupported tile (HWo=%d Cout=%d)", s.name.c_str(), a.HWo, a.Cout); return -1; }
        if (a.G > 0) {
            const int pairs = a.S * a.G;
            if (a.Cv != Cin || Cin % a.G != 0 || pairs > 256 || (pairs & (pairs - 1)) != 0) { *err = fail("conv %s: unsupported GroupNorm shape C=%d", s.name.c_str(), Cin); return -1; }
        }
        if (s.CA % 4 != 0 && s.CB != 0) { *err = fail("conv %s: unaligned concat", s.name.c_str()); return -1; }
        if (conv_lds_bytes(a) > 160 * 1024) { *err = fail("conv %s: LDS tile %zu B too large", s.name.c_str(), conv_lds_bytes(a)); return -1; }
        // tap table: row m = s*HWo + (oy*Wo + ox) -> LDS row s*HWv + iy*Wv + ix, or the zero row
        const int zrow = a.S * a.HWv, zrow_sc = a.S * a.HWo;
        op.tab.assign((size_t)a.Mpad * 10, zrow);
        for (int m = 0; m < a.Mpad; ++m) {
            const int sidx = m / a.HWo, p = m % a.HWo;
            const bool real = m < a.S * a.HWo;
            const int oy = p / s.Wo, ox = p % s.Wo;
            for (int t = 0; t < 9; ++t) {
                const int iy = oy * s.stride + t / 3 - s.pad_lo, ix = ox * s.stride + t % 3 - s.pad_lo;
                if (real && iy >= 0 && iy < s.Hv && ix >= 0 && ix < s.Wv) op.tab[(size_t)m * 10 + t] = sidx * a.HWv + iy * s.Wv + ix;
            }
            op.tab[(size_t)m * 10 + 9] = real ? m : zrow_sc;
        }
        if (s.Ha != s.Hv || s.Wa != s.Wv) { op.mapA = nearest_map(s.Ha, s.Wa, s.Hv, s.Wv); op.has_mapA = true; }
        if (a.Csc && (s.Hsa != s.Ho || s.Wsa != s.Wo)) { op.mapSc = nearest_map(s.Hsa, s.Wsa, s.Ho, s.Wo); op.has_mapSc = true; }
        op.tab_off = ints.size(); ints.insert(ints.end(), op.tab.begin(), op.tab.end());
        if (op.has_mapA) { op.mapA_off = ints.size(); ints.insert(ints.end(), op.mapA.begin(), op.mapA.end()); }
        if (op.has_mapSc) { op.mapSc_off = ints.size(); ints.insert(ints.end(), op.mapSc.begin(), op.mapSc.end()); }
        // weights
        op.w_off = pack_conv(s.conv, Cin, s.Cout);
        op.p_bias = s.conv + ".bias";
        if (!s.gn.empty()) { op.p_gamma = s.gn + ".weight"; op.p_beta = s.gn + ".bias"; }
        if (a.Csc) { op.wsc_off = pack_nin(s.nin, s.CscA + s.CscB, s.Cout); op.p_bias_sc = s.nin + ".b"; }
        op.flops_per_sample = 2.0 * a.HWo * s.Cout * (9.0 * Cin + (s.CscA + s.CscB));
        use(s.tA); use(s.tB); use(s.tScA); use(s.tScB); use(s.tRes);
        int out = -1;
        if (!s.to_output) { out = new_tensor(s.name, s.Cout, s.Ho, s.Wo); }
        op.out_tensor = out;
        c->ops.push_back(op);
        return out;
    }

    int add_attn(const std::string& name, int tin, int C, int H, int W, int* err) {
        Op op;
        op.kind = OP_ATTN;
        op.name = name;
        if (C != 64 || H * W > 96) { *err = fail("attention %s: only C=64, H*W<=96 is built (got C=%d, L=%d)", name.c_str(), C, H * W); return -1; }
        AttnArgs& a = op.attn;
        a.L = H * W; a.Lpad = pad16(a.L); a.G = std::min(C / 4, 32);
        a.eps = 1e-6f; a.scale = 1.0f / std::sqrt((float)C); a.out_scale = (float)(1.0 / std::sqrt(2.0));
        op.tA = tin;
        op.w_off = alloc_w((size_t)3 * C * C);
        for (int i = 0; i < 3; ++i) {
            PackJob j{};
            j.dst = reinterpret_cast<float*>(op.w_off + (size_t)i * C * C);
            j.Cin = C; j.Cout = C; j.Kpad = C; j.Npad = C; j.n_off = 0; j.ntap = 1; j.s_co = 1; j.s_ci = C; j.s_t = 0; j.kind = 0;
            c->jobs.push_back(j);
            c->job_param.push_back(c->pindex.at(name + ".NIN_" + std::to_string(i) + ".W"));
        }
        c->wmap[name + ".qkv"] = op.w_off;
        {   // fused-plan packing: q|k|v side by side as one [C/16][3C][16] matrix
            const size_t o3 = alloc_w((size_t)3 * C * C);
            for (int i = 0; i < 3; ++i) {
                PackJob j{};
                j.dst = reinterpret_cast<float*>(o3);
                j.Cin = C; j.Cout = C; j.Kpad = C; j.Npad = 3 * C; j.n_off = i * C; j.ntap = 1; j.s_co = 1; j.s_ci = C; j.s_t = 0; j.kind = 0;
                c->jobs.push_back(j);
                c->job_param.push_back(c->pindex.at(name + ".NIN_" + std::to_string(i) + ".W"));
            }
            c->wmap[name + ".qkv3"] = o3;
        }
        {   // the same block with NIN_3 folded into the value projection: softmax(QK^T) (V W3) == (softmax(QK^T) V) W3, so the inference
            // programs of the workgroup-resident kernel run no NIN_3 op at all (fused_attn); v-columns = NIN_2.W x NIN_3.W, v-bias = NIN_2.b x NIN_3.W
            const size_t o3 = alloc_w((size_t)3 * C * C);
            for (int i = 0; i < 3; ++i) {
                PackJob j{};
                j.dst = reinterpret_cast<float*>(o3);
                j.Cin = C; j.Cout = C; j.Kpad = C; j.Npad = 3 * C; j.n_off = i * C; j.ntap = 1; j.s_co = 1; j.s_ci = C; j.s_t = 0; j.kind = i == 2 ? 3 : 0;
                c->jobs.push_back(j);
                c->job_param.push_back(c->pindex.at(name + ".NIN_" + std::to_string(i) + ".W"));
                c->job_param2.resize(c->jobs.size(), -1);
                if (i == 2) c->job_param2.back() = c->pindex.at(name + ".NIN_3.W");
            }
            c->wmap[name + ".qkv3f"] = o3;
            const size_t ob = alloc_w((size_t)3 * C);
            c->wmap[name + ".bqkvf"] = ob;
            for (int i = 0; i < 2; ++i) job_copy(name + ".NIN_" + std::to_string(i) + ".b", ob, C, i * C);
            PackJob j{};
            j.dst = reinterpret_cast<float*>(ob);
            j.Cin = C; j.Cout = C; j.n_off = 2 * C; j.kind = 4;
            c->jobs.push_back(j);
            c->job_param.push_back(c->pindex.at(name + ".NIN_2.b"));
            c->job_param2.resize(c->jobs.size(), -1);
            c->job_param2.back() = c->pindex.at(name + ".NIN_3.W");
        }
        op.bqkv_off = alloc_w((size_t)3 * C);
        c->wmap[name + ".bqkv"] = op.bqkv_off;
        for (int i = 0; i < 3; ++i) job_copy(name + ".NIN_" + std::to_string(i) + ".b", op.bqkv_off, C, i * C);
        op.w3_off = pack_nin(name + ".NIN_3", C, C);
        op.p_b3 = name + ".NIN_3.b";
        op.p_gamma = name + ".GroupNorm_0.weight"; op.p_beta = name + ".GroupNorm_0.bias";
        op.flops_per_sample = 2.0 * (4.0 * a.L * C * C + 2.0 * a.L * a.L * C);
        op.attn_C = C; op.attn_H = H; op.attn_W = W;
        use(tin);
        const int out = new_tensor(name, C, H, W);
        op.out_tensor = out;
        c->ops.push_back(op);
        return out;
    }
};

// one ResnetBlockDDPMpp = two fused conv launches
int add_resblock(Builder& b, const std::string& name, int tA, int tB, int CA, int CB, int Ha, int Wa, int H, int W,
                 int cout, int dense_off, int* err) {
    ConvSpec s0;
    s0.name = name + ".Conv_0";
    s0.tA = tA; s0.tB = tB; s0.CA = CA; s0.CB = CB; s0.Ha = Ha; s0.Wa = Wa; s0.Hv = H; s0.Wv = W;
    s0.gn = name + ".GroupNorm_0"; s0.conv = name + ".Conv_0";
    s0.Ho = H; s0.Wo = W; s0.Cout = cout; s0.dense_off = dense_off;
    const int h1 = b.add_conv(s0, err);
    if (*err) return -1;
    ConvSpec s1;
    s1.name = name;
    s1.tA = h1; s1.CA = cout; s1.Ha = H; s1.Wa = W; s1.Hv = H; s1.Wv = W;
    s1.gn = name + ".GroupNorm_1"; s1.conv = name + ".Conv_1";
    s1.Ho = H; s1.Wo = W; s1.Cout = cout;
    s1.scale = (float)(1.0 / std::sqrt(2.0));
    s1.dropout = true;
    if (CA + CB != cout) {
        s1.tScA = tA; s1.tScB = tB; s1.CscA = CA; s1.CscB = CB; s1.Hsa = Ha; s1.Wsa = Wa; s1.nin = name + ".NIN_0";
    } else {
        if (tB >= 0 || Ha != H || Wa != W) { *err = fail("%s: identity shortcut over a gathered input is not built", name.c_str()); return -1; }
        s1.tRes = tA;
    }
    return b.add_conv(s1, err);
}


// ------------------------------------------------------------------------------------------
// tiled plan (csrc/tiled_kernels.h): NCSNpp.forward (RD/models/ncsnpp.py:226-354) as a list of launches over
// HBM-resident NHWC tensors, for shapes beyond one workgroup per sample (the CIFAR-shape model of BASELINE config #5)
// ------------------------------------------------------------------------------------------
struct TiledBuilder {
    rdmi_ctx* c; Builder& b;
    size_t top = 0;                                   // floats per sample allocated so far
    struct TT { size_t off = 0; int C = 0, H = 0, W = 0; bool valid = false; size_t cs = (size_t)-1; int tiles = 0, tile_px = 0; bool bf = false; bool raw16 = false; };   // raw16: q | k | v stored as bf16 [HW][3C]: only the fused attention core and its V transpose read it   // cs: per-tile channel sums of the producer; bf: a bf16 [HW][C] tensor (C % 64 == 0) for tconv_pre
    TT talloc(int C, int H, int W) { TT t; t.off = top; t.C = C; t.H = H; t.W = W; t.valid = true; top += ((size_t)C * H * W + 63) & ~(size_t)63; return t; }
    static int pad32(int a) { return (a + 31) & ~31; }

    bool bf16() const { return c->arch.compute_dtype == 1; }
    // weights: fp32 [tap][K/16][Np][16], or the bf16 copy [tap][K/32][Np][32] (half the arena floats)
    size_t pack3x3(const std::string& pre, int cin, int cout) {
        const int Kp = pad32(cin), Np = pad16(cout);
        const size_t o = b.alloc_w(bf16() ? ((size_t)9 * Kp * Np + 1) / 2 : (size_t)9 * Kp * Np);
        b.job_pack(pre + ".weight", o, cin, cout, Kp, Np, 0, 9, (long)cin * 9, 9, 1);
        if (bf16()) c->jobs.back().kind = 2;
        return o;
    }
    size_t pack1x1(const std::string& pre, int cin, int cout) {
        const int Kp = pad32(cin), Np = pad16(cout);
        const size_t o = b.alloc_w(bf16() ? ((size_t)Kp * Np + 1) / 2 : (size_t)Kp * Np);
        b.job_pack(pre + ".W", o, cin, cout, Kp, Np, 0, 1, 1, cout, 0);
        if (bf16()) c->jobs.back().kind = 2;
        return o;
    }
    TT stats(const std::string& name, const TT& A, const TT* B) {
        const int C = A.C + (B ? B->C : 0), G = std::min(C / 4, 32);
        TT st = talloc(2 * G, 1, 1);
        if (A.cs != (size_t)-1 && (!B || B->cs != (size_t)-1) && std::getenv("RDMI_TILED_STATS_PASS") == nullptr) {
            // statistics from the channel sums the producing convs left behind: no pass over the tensors
            rdmi_ctx::TLaunch f; f.kind = 5; f.name = name + ".stats";
            f.oA = A.cs; f.oB = B ? B->cs : rdmi_ctx::TLaunch::NONE; f.CA = A.C; f.CB = B ? B->C : 0; f.HW = A.H * A.W; f.G = G;
            f.tL = A.tiles; f.tC = B ? B->tiles : 0; f.pxA = A.tile_px; f.pxB = B ? B->tile_px : 0; f.oStats = st.off;
            c->tl.push_back(f);
            return st;
        }
        rdmi_ctx::TLaunch l; l.kind = 1; l.name = name + ".stats";
        l.oA = A.off; l.oB = B ? B->off : rdmi_ctx::TLaunch::NONE; l.CA = A.C; l.CB = B ? B->C : 0; l.HW = A.H * A.W; l.G = G;
        l.oStats = st.off;
        c->tl.push_back(l);
        return st;
    }
    // conv over concat(A, B) [optionally GroupNorm(+SiLU)'d with `st`], 3x3 (stride 1 pad 1 | stride 2 Downsample | nearest x2 Upsample) or 1x1
    TT conv(const std::string& name, const TT& A, const TT* B, const TT* st, const std::string& gn, bool act, int ntap, int stride, bool up,
            size_t w_off, int cout, const std::string& bias_param, size_t bias_arena, int dense_off, const TT* resid, float scale, bool final_out,
            TT* raw_copy = nullptr, bool out16 = false) {       // out16: the output is stored as bf16 (TT::raw16)     // raw_copy: if the conv takes the pre-activated form, its activation pass also leaves the RAW input as a bf16 [HW][Cv] tensor here
        // bf16 plan: a 3x3 stride-1 conv behind a GroupNorm reads a tensor that was normalised, activated and rounded to bf16 ONCE
        // (kind 6) instead of redoing that arithmetic for every staged window element (RDMI_NO_PREACT=1: the one-kernel form)
        // (3x3 stride-1 convs and the 1x1 q/k/v projection of attention blocks); a tensor that already is bf16 (the fused attention
        // core's output) goes to the same kernel without the extra pass.
        const bool direct = A.bf && !B && !st && ntap == 1;
        const bool pre = direct || (bf16() && st && (ntap == 9 || ntap == 1) && stride == 1 && !up && !final_out && cout % 16 == 0 && A.C % 4 == 0 &&
                                    (!B || B->C % 4 == 0) && (A.C + (B ? B->C : 0)) % 64 == 0 && std::getenv("RDMI_NO_PREACT") == nullptr);
        if (A.bf && !direct) throw std::runtime_error("tiled plan: a bf16 tensor feeds a conv that cannot take it (" + name + ")");
        if (A.raw16 || (B && B->raw16) || (resid && resid->raw16)) throw std::runtime_error("tiled plan: a bf16-stored q | k | v tensor is read by a conv (" + name + ")");
        size_t act_off = A.off;
        if (pre && !direct) {
            const int Cin = A.C + (B ? B->C : 0), Cvp = pad32(Cin);
            TT actT = talloc((Cvp + 1) / 2, A.H, A.W);               // bf16 [HW][Cv]
            act_off = actT.off;
            rdmi_ctx::TLaunch g; g.kind = 6; g.name = name + ".act";
            g.oA = A.off; g.oB = B ? B->off : rdmi_ctx::TLaunch::NONE; g.oStats = st->off; g.oOut = actT.off;
            if (raw_copy && std::getenv("RDMI_NO_RAW_COPY") == nullptr) {
                *raw_copy = talloc((Cvp + 1) / 2, A.H, A.W);
                raw_copy->C = Cvp; raw_copy->bf = true;
                g.oOut2 = raw_copy->off;
            }
            g.gact.CA = A.C; g.gact.CB = B ? B->C : 0; g.gact.Cv = Cvp; g.gact.HW = A.H * A.W;
            g.gact.G = std::min(Cin / 4, 32); g.gact.Cg = Cin / g.gact.G; g.gact.act = act ? 1 : 0;
            g.p_gamma = gn + ".weight"; g.p_beta = gn + ".bias";
            // the statistics launch of this GroupNorm (kind 5: from the producers' channel records) is folded into the activation pass:
            // gn_act_fin_kernel reads the records itself (RDMI_NO_GN_FOLD=1 keeps the two launches)
            int max_slots = 0;                                          // groups a 64-channel slice touches (the kernel's table holds 16)
            for (int c_lo = 0; c_lo < Cin; c_lo += 64) max_slots = std::max(max_slots, (std::min(c_lo + 64, Cin) - 1) / g.gact.Cg - c_lo / g.gact.Cg + 1);
            if (Cvp % 64 == 0 && max_slots <= 16 && g.gact.Cg <= 16 && std::getenv("RDMI_NO_GN_FOLD") == nullptr)
                for (size_t k = c->tl.size(); k-- > 0 && k + 4 > c->tl.size();)
                    if (c->tl[k].kind == 5 && c->tl[k].oStats == st->off) {
                        const rdmi_ctx::TLaunch& f = c->tl[k];
                        g.fin = true; g.oCsA = f.oA; g.oCsB = f.oB;
                        g.gact.tilesA = f.tL; g.gact.tilesB = f.tC; g.gact.pxA = f.pxA; g.gact.pxB = f.pxB; g.gact.eps = 1e-6f;
                        c->tl.erase(c->tl.begin() + (long)k);
                        break;
                    }
            c->tl.push_back(g);
        }
        TConvArgs a{};
        a.CA = A.C; a.CB = B ? B->C : 0; a.Cv = pad32(a.CA + a.CB);
        if (pre) { a.CA = a.Cv; a.CB = 0; }
        a.Ha = A.H; a.Wa = A.W; a.up = up ? 1 : 0; a.Hv = up ? 2 * A.H : A.H; a.Wv = up ? 2 * A.W : A.W;
        a.stride = stride; a.ntap = ntap; a.pad_lo = (ntap == 9 && stride == 1) ? 1 : 0;
        a.Ho = stride == 2 ? (a.Hv + 1 - 3) / 2 + 1 : a.Hv; a.Wo = stride == 2 ? (a.Wv + 1 - 3) / 2 + 1 : a.Wv;
        a.TR = a.Wo >= 64 ? 1 : std::max(1, std::min(a.Ho, 64 / a.Wo));
        if (st && !pre) { a.G = std::min((a.CA + a.CB) / 4, 32); a.Cg = (a.CA + a.CB) / a.G; a.act = act ? 1 : 0; }
        a.dense_off = dense_off < 0 ? 0 : dense_off; a.dense_stride = c->dense_total;
        a.out_scale = scale;
        a.Cout = cout; a.Cout_pad = pad16(cout);
        // bf16 packs: interleave the weights' columns by the NCT the launch will use at the sampling batch (run_tiled's rule at 128 forwards),
        // so that the epilogue's loads and stores are 8- / 16-byte vectors over full lines (tconv_epilogue; RDMI_NO_COL_IL=1: plain order)
        if (bf16() && std::getenv("RDMI_NO_COL_IL") == nullptr) {
            const long tl = ceil_div(a.Ho, a.TR);
            for (int cand : {4, 2})
                if (cout % (16 * cand) == 0 && a.Cout_pad >= 64 * cand && tl * 128 * ceil_div(a.Cout_pad, 64 * cand) >= 512) { a.col_il = cand; break; }
            if (a.col_il > 1)
                for (auto& j : c->jobs) if (j.kind == 2 && j.dst == reinterpret_cast<float*>(w_off)) j.col_il = a.col_il;
        }
        TT out = talloc(out16 ? (cout + 1) / 2 : cout, a.Ho, a.Wo);
        out.C = cout; out.raw16 = out16; a.out_bf16 = out16 ? 1 : 0;
        size_t cs_off = rdmi_ctx::TLaunch::NONE;
        if (!final_out && cout % 4 == 0) {
            out.tiles = ceil_div(a.Ho, a.TR); out.tile_px = a.TR * a.Wo;
            TT cs = talloc(2 * cout * out.tiles, 1, 1);
            out.cs = cs_off = cs.off;
        }
        rdmi_ctx::TLaunch l; l.kind = 0; l.name = name; l.conv = a;
        l.oC = cs_off;
        l.oA = A.off; l.oB = B ? B->off : rdmi_ctx::TLaunch::NONE; l.oStats = st ? st->off : rdmi_ctx::TLaunch::NONE;
        if (pre) { l.pre = true; l.oA = act_off; l.oB = rdmi_ctx::TLaunch::NONE; l.oStats = rdmi_ctx::TLaunch::NONE; }
        l.oResid = resid ? resid->off : rdmi_ctx::TLaunch::NONE; l.oOut = out.off;
        l.nmt = (a.TR * a.Wo > 16) ? 4 : 1;
        l.w_off = w_off; l.p_bias = bias_param; l.bias_arena = bias_arena;
        if (st && !pre) { l.p_gamma = gn + ".weight"; l.p_beta = gn + ".bias"; }
        l.out_is_final = final_out;
        l.use_dense = dense_off >= 0;
        l.flops_per_sample = 2.0 * a.Ho * a.Wo * cout * (double)ntap * (A.C + (B ? B->C : 0));
        c->tl.push_back(l);
        return out;
    }
    TT resblock(const std::string& name, const TT& A, const TT* B, int cout, int dense_off) {
        const int cin = A.C + (B ? B->C : 0);
        const float rs2 = (float)(1.0 / std::sqrt(2.0));
        TT st0 = stats(name + ".GroupNorm_0", A, B);
        // bf16 plan: GroupNorm_0's activation pass reads the block's raw input anyway and leaves a bf16 copy of it for the NIN_0
        // shortcut, which then is a copy-staged 1x1 conv over half the bytes instead of a conv that converts fp32 while staging
        TT xraw;
        TT h1 = conv(name + ".Conv_0", A, B, &st0, name + ".GroupNorm_0", true, 9, 1, false, pack3x3(name + ".Conv_0", cin, cout), cout,
                     name + ".Conv_0.bias", (size_t)-1, dense_off, nullptr, 1.f, false, cin != cout ? &xraw : nullptr);
        TT st1 = stats(name + ".GroupNorm_1", h1, nullptr);
        const size_t w1 = pack3x3(name + ".Conv_1", cout, cout);
        if (cin != cout) {
            TT sc = xraw.valid ? conv(name + ".NIN_0", xraw, nullptr, nullptr, "", false, 1, 1, false, pack1x1(name + ".NIN_0", cin, cout), cout, name + ".NIN_0.b", (size_t)-1, -1,
                                      nullptr, 1.f, false)
                               : conv(name + ".NIN_0", A, B, nullptr, "", false, 1, 1, false, pack1x1(name + ".NIN_0", cin, cout), cout, name + ".NIN_0.b", (size_t)-1, -1,
                                      nullptr, 1.f, false);
            return conv(name + ".Conv_1", h1, nullptr, &st1, name + ".GroupNorm_1", true, 9, 1, false, w1, cout, name + ".Conv_1.bias", (size_t)-1, -1, &sc, rs2, false);
        }
        return conv(name + ".Conv_1", h1, nullptr, &st1, name + ".GroupNorm_1", true, 9, 1, false, w1, cout, name + ".Conv_1.bias", (size_t)-1, -1, &A, rs2, false);
    }
    // AttnBlockpp (RD/models/layerspp.py:67-96): GN -> q,k,v (one 1x1 conv, Cout = 3C) -> softmax(q k^T / sqrt(C)) v -> NIN_3 -> (x + h)/sqrt2
    TT attn(const std::string& name, const TT& x) {
        const int C = x.C, Lq = x.H * x.W;
        if (C % 32 != 0 || Lq % 16 != 0) throw std::runtime_error("tiled attention needs C % 32 == 0 and H*W % 16 == 0");
        TT st = stats(name + ".GroupNorm_0", x, nullptr);
        const size_t o3 = b.alloc_w(bf16() ? (size_t)3 * C * C / 2 : (size_t)3 * C * C);
        for (int i = 0; i < 3; ++i) {
            PackJob j{};
            j.dst = reinterpret_cast<float*>(o3);
            j.Cin = C; j.Cout = C; j.Kpad = C; j.Npad = 3 * C; j.n_off = i * C; j.ntap = 1; j.s_co = 1; j.s_ci = C; j.s_t = 0; j.kind = bf16() ? 2 : 0;
            c->jobs.push_back(j);
            c->job_param.push_back(c->pindex.at(name + ".NIN_" + std::to_string(i) + ".W"));
        }
        const size_t bq = b.alloc_w((size_t)3 * C);
        for (int i = 0; i < 3; ++i) b.job_copy(name + ".NIN_" + std::to_string(i) + ".b", bq, C, i * C);
        const bool flash = bf16() && (C == 64 || C == 128 || C == 256) && Lq % 64 == 0 && std::getenv("RDMI_NO_FLASH") == nullptr;
        // the fused core rounds q, k, v to bf16 before its MFMAs anyway: the projection stores them as bf16 (half the conv's writes, half the
        // core's and the transpose's reads; same bits into the MFMAs when 1 / sqrt(C) is a power of two, C = 64 / 256).  RDMI_NO_QKV16=1: fp32
        const bool qkv16 = flash && C % 2 == 0 && std::getenv("RDMI_NO_QKV16") == nullptr;
        TT qkv = conv(name + ".qkv", x, nullptr, &st, name + ".GroupNorm_0", false, 1, 1, false, o3, 3 * C, "", bq, -1, nullptr, 1.f, false, nullptr, qkv16);
        TT Vt = talloc(qkv16 ? C / 2 : C, Lq, 1);    // [C][L]
        TT O = talloc(C, x.H, x.W);
        if (flash) { O.bf = true; }                  // written as bf16 [L][C] (half of the allocation): NIN_3 stages plain copies of it
        if (flash) {
            // bf16 plan: scores, softmax and P V in one kernel (no [L][L] buffer); V^T still comes from the transpose launch
            { rdmi_ctx::TLaunch l; l.kind = 4; l.name = name + ".vT"; l.oA = qkv.off; l.oOut = Vt.off; l.tL = Lq; l.tC = C; l.tld = 3 * C; l.tc0 = 2 * C; l.t16 = qkv16; c->tl.push_back(l); }
            rdmi_ctx::TLaunch l; l.kind = 7; l.name = name + ".core"; l.oA = qkv.off; l.oB = Vt.off; l.oOut = O.off;
            l.flash.L = Lq; l.flash.alpha = 1.0f / std::sqrt((float)C); l.flashC = C; l.flash.out_bf16 = 1; l.flash.in_bf16 = qkv16 ? 1 : 0;
            l.flops_per_sample = 4.0 * Lq * Lq * C;
            c->tl.push_back(l);
        } else {
        TT S = talloc(Lq, Lq, 1);                    // [L][L] scores / probabilities
        {
            rdmi_ctx::TLaunch l; l.kind = 2; l.name = name + ".qk";
            l.oA = qkv.off; l.oB = qkv.off; l.dB = C; l.oC = S.off;
            l.gemm.sa = l.gemm.sb = (long)Lq * 3 * C; l.gemm.sc = (long)Lq * Lq; l.gemm.lda = l.gemm.ldb = 3 * C; l.gemm.ldc = Lq;
            l.gemm.M = Lq; l.gemm.N = Lq; l.gemm.K = C; l.gemm.alpha = 1.0f / std::sqrt((float)C);
            l.flops_per_sample = 2.0 * Lq * Lq * C;
            c->tl.push_back(l);
        }
        { rdmi_ctx::TLaunch l; l.kind = 3; l.name = name + ".softmax"; l.oA = S.off; l.rows_per_sample = Lq; l.L = Lq; c->tl.push_back(l); }
        { rdmi_ctx::TLaunch l; l.kind = 4; l.name = name + ".vT"; l.oA = qkv.off; l.oOut = Vt.off; l.tL = Lq; l.tC = C; l.tld = 3 * C; l.tc0 = 2 * C; c->tl.push_back(l); }
        {
            rdmi_ctx::TLaunch l; l.kind = 2; l.name = name + ".pv";
            l.oA = S.off; l.oB = Vt.off; l.oC = O.off;
            l.gemm.sa = (long)Lq * Lq; l.gemm.sb = (long)C * Lq; l.gemm.sc = (long)Lq * C; l.gemm.lda = Lq; l.gemm.ldb = Lq; l.gemm.ldc = C;
            l.gemm.M = Lq; l.gemm.N = C; l.gemm.K = Lq; l.gemm.alpha = 1.f;
            l.flops_per_sample = 2.0 * Lq * Lq * C;
            c->tl.push_back(l);
        }
        }
        return conv(name + ".NIN_3", O, nullptr, nullptr, "", false, 1, 1, false, pack1x1(name + ".NIN_3", C, C), C, name + ".NIN_3.b", (size_t)-1, -1, &x,
                    (float)(1.0 / std::sqrt(2.0)), false);
    }
};

int build_tiled_plan(rdmi_ctx* c, Builder& b, const Layout& L, std::map<std::string, int>& dense_off) {
    const rdmi_arch& a = c->arch;
    using TT = TiledBuilder::TT;
    TiledBuilder t{c, b};
    int H = c->H, W = c->W;
    TT xin = t.talloc(a.channels, H, W);               // NHWC copy of the caller's NCHW input
    xin.off = rdmi_ctx::TLaunch::XIN;                  // marker: resolved to t_xin
    TT h = t.conv("input_conv", xin, nullptr, nullptr, "", false, 9, 1, false, t.pack3x3("input_conv", a.channels, a.nf), a.nf, "input_conv.bias", (size_t)-1, -1,
                  nullptr, 1.f, false);
    c->tl.back().in_is_x = true;
    std::vector<TT> hs{h};
    int d = 0;
    for (int i = 0; i < a.n_levels; ++i) {
        for (int j = 0; j < a.num_res_blocks; ++j, ++d) {
            const BlockSpec& bs = L.down[(size_t)d];
            h = t.resblock(bs.name, h, nullptr, bs.cout, dense_off[bs.name]);
            if (bs.attn) h = t.attn("down_attn." + std::to_string(d), h);
            hs.push_back(h);
        }
        hs.push_back(h);
        if (i != a.n_levels - 1) {
            const std::string nm = "downsample." + std::to_string(i) + ".Conv_0";
            h = t.conv(nm, h, nullptr, nullptr, "", false, 9, 2, false, t.pack3x3(nm, h.C, h.C), h.C, nm + ".bias", (size_t)-1, -1, nullptr, 1.f, false);
        }
    }
    h = t.resblock("mid_block1", h, nullptr, L.mid_ch, dense_off["mid_block1"]);
    if (a.attn_levels >> (a.n_levels - 1) & 1) return fail("attention at the bottleneck resolution (mid_attn) is not built");
    h = t.resblock("mid_block2", h, nullptr, L.mid_ch, dense_off["mid_block2"]);
    int u = 0;
    for (int k = 0; k < a.n_levels; ++k) {
        for (int j = 0; j < a.num_res_blocks + 1; ++j, ++u) {
            const BlockSpec& bs = L.up[(size_t)u];
            TT sk = hs.back();
            hs.pop_back();
            if (sk.H != h.H || sk.W != h.W) return fail("tiled plan: skip grid %dx%d != %dx%d (the nearest-resize fix of odd grids is only built for the workgroup-resident plan)", sk.H, sk.W, h.H, h.W);
            h = t.resblock(bs.name, h, &sk, bs.cout, dense_off[bs.name]);
            if (bs.attn) h = t.attn("up_attn." + std::to_string(u), h);
        }
        if (k != a.n_levels - 1) {
            const std::string nm = "upsample." + std::to_string(k) + ".Conv_0";
            h = t.conv(nm, h, nullptr, nullptr, "", false, 9, 1, true, t.pack3x3(nm, h.C, h.C), h.C, nm + ".bias", (size_t)-1, -1, nullptr, 1.f, false);
        }
    }
    if (h.H != c->H || h.W != c->W) return fail("network output grid %dx%d != input %dx%d", h.H, h.W, c->H, c->W);
    TT st = t.stats("out_norm", h, nullptr);
    t.conv("out_conv", h, nullptr, &st, "out_norm", true, 9, 1, false, t.pack3x3("out_conv", h.C, a.channels), a.channels, "out_conv.bias", (size_t)-1, -1, nullptr, 1.f, true);
    c->t_ws_per_sample = t.top;
    c->tiled = true;
    return 0;
}

// resolve a per-sample workspace offset to a device pointer (tensors are [tensor][n][...]: the offset scales with max_batch)
inline float* tl_ptr(rdmi_ctx* c, size_t off, long delta = 0) {
    if (off == rdmi_ctx::TLaunch::NONE) return nullptr;
    if (off == rdmi_ctx::TLaunch::XIN) return c->t_xin;
    return c->t_ws + off * (size_t)c->max_batch + delta;
}

int finish_tiled_plan(rdmi_ctx* c) {
    const size_t NBmax = (size_t)c->max_batch;
    const size_t E = (size_t)c->H * c->W * c->arch.channels;
    HIP_OK(hipMalloc((void**)&c->t_ws, c->t_ws_per_sample * NBmax * sizeof(float)));
    HIP_OK(hipMalloc((void**)&c->t_xin, E * NBmax * sizeof(float)));
    HIP_OK(hipMalloc((void**)&c->t_out, E * NBmax * sizeof(float)));
    HIP_OK(hipMalloc(&c->t_zero, 256));
    HIP_OK(hipMemset(c->t_zero, 0, 256));
    // measured on MI355X (rocprofv3 kernel trace, B = 64 with guidance): 5.28 ms for the 3x3 convs of the 32x32 / 16x16 levels against 4.88 ms
    // with tconv_pre_kernel -- the implicit-GEMM form is NOT the default; RDMI_ICONV=1 selects it (from RDMI_ICONV_MIN_WGS tiles, default 256)
    if (const char* e = std::getenv("RDMI_TILED_MIN_WGS")) c->tiled_min_wgs = std::atol(e);
    c->iconv_min_wgs = LONG_MAX;
    if (const char* e = std::getenv("RDMI_ICONV")) if (std::atoi(e) != 0) c->iconv_min_wgs = 256;
    if (const char* e = std::getenv("RDMI_ICONV_MIN_WGS")) c->iconv_min_wgs = std::atol(e);
    for (auto& l : c->tl) {
        if (l.kind == 0) {
            TConvArgs& a = l.conv;
            a.srcA = tl_ptr(c, l.oA); a.srcB = tl_ptr(c, l.oB); a.stats = tl_ptr(c, l.oStats); a.resid = tl_ptr(c, l.oResid);
            a.out = l.out_is_final ? c->t_out : tl_ptr(c, l.oOut);
            a.wpk = c->d_w + l.w_off;
            a.dense = l.use_dense ? c->d_dense : nullptr;
            a.chsum = tl_ptr(c, l.oC);
            a.zeros = c->t_zero;
            if (l.pre && tconv_trv(a) * tconv_wl(a) * (a.ntap == 1 ? TpCfg<1>::UPP : TpCfg<9>::UPP) > TC_MAXS * RDMI_THREADS)
                return fail("tiled conv %s: window of %d pixels exceeds the register staging", l.name.c_str(), tconv_trv(a) * tconv_wl(a));
            if (l.pre && tconv_pre_lds_bytes(a) > 160 * 1024) return fail("tiled conv %s: LDS window %zu B", l.name.c_str(), tconv_pre_lds_bytes(a));
            if (tconv_lds_bytes(a) > 160 * 1024) return fail("tiled conv %s: LDS window %zu B", l.name.c_str(), tconv_lds_bytes(a));
            if (tconv_trv(a) * tconv_wl(a) * 8 > TC_MAXS * RDMI_THREADS) return fail("tiled conv %s: window of %d pixels exceeds the register staging (%d float4 per work-item)", l.name.c_str(), tconv_trv(a) * tconv_wl(a), TC_MAXS);
        } else if (l.kind == 1) {
            l.sA = tl_ptr(c, l.oA); l.sB = tl_ptr(c, l.oB); l.stats = tl_ptr(c, l.oStats);
        } else if (l.kind == 2) {
            l.gemm.A = tl_ptr(c, l.oA, l.dA); l.gemm.B = tl_ptr(c, l.oB, l.dB); l.gemm.C = tl_ptr(c, l.oC);
        } else if (l.kind == 3) {
            l.sm = tl_ptr(c, l.oA);
        } else if (l.kind == 5) {
            l.sA = tl_ptr(c, l.oA); l.sB = tl_ptr(c, l.oB); l.stats = tl_ptr(c, l.oStats);
        } else if (l.kind == 7) {
            l.flash.qkv = tl_ptr(c, l.oA); l.flash.vt = tl_ptr(c, l.oB); l.flash.out = tl_ptr(c, l.oOut);
        } else if (l.kind == 6) {
            l.gact.A = tl_ptr(c, l.oA); l.gact.B = tl_ptr(c, l.oB); l.gact.stats = tl_ptr(c, l.oStats);
            l.gact.out = reinterpret_cast<bf16_t*>(tl_ptr(c, l.oOut));
            l.gact.out2 = reinterpret_cast<bf16_t*>(tl_ptr(c, l.oOut2));
            if (l.fin) { l.gact.csA = tl_ptr(c, l.oCsA); l.gact.csB = tl_ptr(c, l.oCsB); l.gact.stats = nullptr; }
        } else {
            l.tsrc = tl_ptr(c, l.oA); l.tdst = tl_ptr(c, l.oOut);
        }
    }
    return 0;
}

// replay the tiled plan for NB samples (embedding already evaluated into d_dense); x is NCHW, out is NCHW
int run_tiled(rdmi_ctx* c, const float* x, int x_mod, const float* sig, int sig_mod, int sig_is_time, float t_scalar, float smin, float ratio, float* out, int NB,
              const float* dense_rows, hipStream_t s);
int build_fused_program(rdmi_ctx* c);

int build_plan(rdmi_ctx* c) {
    const rdmi_arch& a = c->arch;
    Layout L = build_layout(c);
    build_params(c, L);
    Builder b{c};
    int err = 0;
    // Dense_0 offsets in block order (down, mid1, mid2, up)
    std::vector<std::pair<std::string, int>> dense_blocks;
    for (auto& d : L.down) dense_blocks.push_back({d.name, d.cout});
    dense_blocks.push_back({"mid_block1", L.mid_ch});
    dense_blocks.push_back({"mid_block2", L.mid_ch});
    for (auto& u : L.up) dense_blocks.push_back({u.name, u.cout});
    std::map<std::string, int> dense_off;
    int dt = 0;
    for (auto& d : dense_blocks) { dense_off[d.first] = dt; dt += d.second; }
    c->dense_total = dt;

    // embedding weights: time_mlp.0 [temb][2nf], time_mlp.2 [temb][temb], all Dense_0 side by side
    const int T = c->temb, Np_d = (dt + 63) & ~63, Np_t = (T + 63) & ~63;
    c->w_t0 = b.alloc_w((size_t)pad16(2 * a.nf) * Np_t);
    b.job_pack("time_mlp.0.weight", c->w_t0, 2 * a.nf, T, pad16(2 * a.nf), Np_t, 0, 1, 2 * a.nf, 1, 0);
    c->w_t2 = b.alloc_w((size_t)T * Np_t);
    b.job_pack("time_mlp.2.weight", c->w_t2, T, T, T, Np_t, 0, 1, T, 1, 0);
    c->w_dense = b.alloc_w((size_t)T * Np_d);
    c->b_dense = b.alloc_w((size_t)Np_d);
    for (auto& d : dense_blocks) {
        b.job_pack(d.first + ".Dense_0.weight", c->w_dense, T, d.second, T, Np_d, dense_off[d.first], 1, T, 1, 0);
        b.job_copy(d.first + ".Dense_0.bias", c->b_dense, d.second, dense_off[d.first]);
    }

    // Shapes beyond one workgroup per sample (more than 96 pixels, or more than one image channel: the CIFAR-shape model of
    // BASELINE config #5) run the spatially tiled plan; the GTO-Halo shapes keep the workgroup-resident / layer plans below.
    if (a.compute_dtype != 0 && a.compute_dtype != 1) return fail("compute_dtype=%d (0: fp32, 1: bf16)", a.compute_dtype);
    if (c->H * c->W > 96 || a.channels != 1) {
        try { if (int e = build_tiled_plan(c, b, L, dense_off)) return e; }
        catch (const std::exception& ex) { return fail("tiled plan: %s", ex.what()); }
        c->w_floats = b.wf;
        HIP_OK(hipMalloc((void**)&c->d_w, c->w_floats * sizeof(float)));
        HIP_OK(hipMemset(c->d_w, 0, c->w_floats * sizeof(float)));
        HIP_OK(hipMalloc((void**)&c->d_jobs, c->jobs.size() * sizeof(PackJob)));
        const size_t NBm = (size_t)c->max_batch, Mp_ = (size_t)pad16(c->max_batch), E_ = (size_t)c->H * c->W * a.channels;
        HIP_OK(hipMalloc((void**)&c->d_h1, Mp_ * T * sizeof(float)));
        HIP_OK(hipMalloc((void**)&c->d_temb, Mp_ * T * sizeof(float)));
        HIP_OK(hipMalloc((void**)&c->d_dense, Mp_ * dt * sizeof(float)));
        HIP_OK(hipMalloc((void**)&c->d_s2, NBm * E_ * sizeof(float)));
        HIP_OK(hipMalloc((void**)&c->d_score, NBm * E_ * sizeof(float)));
        HIP_OK(hipMalloc((void**)&c->d_z, NBm * E_ * sizeof(float)));
        HIP_OK(hipMalloc((void**)&c->d_norms, (2 * NBm + 2) * sizeof(float)));
        HIP_OK(hipMalloc((void**)&c->d_tvec, NBm * sizeof(float)));
        HIP_OK(hipMalloc((void**)&c->d_state, sizeof(StepState)));
        HIP_OK(hipMemset(c->d_state, 0, sizeof(StepState)));
        for (auto& j : c->jobs) j.dst = c->d_w + reinterpret_cast<size_t>(j.dst);
        return finish_tiled_plan(c);
    }
    // ---- NCSNpp.forward data flow
    int H = c->H, W = c->W;
    ConvSpec in;
    in.name = "input_conv"; in.tA = -2; in.CA = a.channels; in.Ha = H; in.Wa = W; in.Hv = H; in.Wv = W;
    in.conv = "input_conv"; in.Ho = H; in.Wo = W; in.Cout = a.nf;
    int h = b.add_conv(in, &err);
    if (err) return err;
    struct HS { int t, C, H, W; };
    std::vector<HS> hs{{h, a.nf, H, W}};
    int ch = a.nf, d = 0;
    for (int i = 0; i < a.n_levels; ++i) {
        for (int j = 0; j < a.num_res_blocks; ++j, ++d) {
            const BlockSpec& bs = L.down[d];
            h = add_resblock(b, bs.name, h, -1, bs.cin, 0, H, W, H, W, bs.cout, dense_off[bs.name], &err);
            if (err) return err;
            ch = bs.cout;
            if (bs.attn) { h = b.add_attn("down_attn." + std::to_string(d), h, ch, H, W, &err); if (err) return err; }
            hs.push_back({h, ch, H, W});
        }
        hs.push_back({h, ch, H, W});
        if (i != a.n_levels - 1) {
            ConvSpec s;
            s.name = "downsample." + std::to_string(i);
            s.tA = h; s.CA = ch; s.Ha = H; s.Wa = W; s.Hv = H; s.Wv = W;
            s.conv = s.name + ".Conv_0"; s.stride = 2; s.pad_lo = 0;
            s.Ho = (H + 1 - 3) / 2 + 1; s.Wo = (W + 1 - 3) / 2 + 1; s.Cout = ch;
            h = b.add_conv(s, &err);
            if (err) return err;
            H = s.Ho; W = s.Wo;
        }
    }
    h = add_resblock(b, "mid_block1", h, -1, ch, 0, H, W, H, W, ch, dense_off["mid_block1"], &err);
    if (err) return err;
    h = add_resblock(b, "mid_block2", h, -1, ch, 0, H, W, H, W, ch, dense_off["mid_block2"], &err);
    if (err) return err;
    int u = 0;
    for (int k = 0; k < a.n_levels; ++k) {
        for (int j = 0; j < a.num_res_blocks + 1; ++j, ++u) {
            const BlockSpec& bs = L.up[u];
            HS sk = hs.back();
            hs.pop_back();
            // h is gathered (nearest) onto the skip's grid when the shapes differ (RD/models/ncsnpp.py:319-320)
            const int Hs = H, Ws = W;
            H = sk.H; W = sk.W;
            h = add_resblock(b, bs.name, h, sk.t, ch, sk.C, Hs, Ws, H, W, bs.cout, dense_off[bs.name], &err);
            if (err) return err;
            ch = bs.cout;
            if (bs.attn) { h = b.add_attn("up_attn." + std::to_string(u), h, ch, H, W, &err); if (err) return err; }
        }
        if (k != a.n_levels - 1) {
            ConvSpec s;
            s.name = "upsample." + std::to_string(k);
            s.tA = h; s.CA = ch; s.Ha = H; s.Wa = W; s.Hv = 2 * H; s.Wv = 2 * W;
            s.conv = s.name + ".Conv_0"; s.Ho = 2 * H; s.Wo = 2 * W; s.Cout = ch;
            h = b.add_conv(s, &err);
            if (err) return err;
            H *= 2; W *= 2;
        }
    }
    if (H != c->H || W != c->W) return fail("network output grid %dx%d != input %dx%d", H, W, c->H, c->W);
    ConvSpec out;
    out.name = "out_conv"; out.tA = h; out.CA = ch; out.Ha = H; out.Wa = W; out.Hv = H; out.Wv = W;
    out.gn = "out_norm"; out.conv = "out_conv"; out.Ho = H; out.Wo = W; out.Cout = a.channels; out.to_output = true;
    b.add_conv(out, &err);
    if (err) return err;

    // ---- liveness-packed workspace offsets
    {
        struct Blk { size_t off, size; };
        std::vector<Blk> freel;
        std::vector<std::vector<int>> dies((size_t)c->ops.size() + 1);
        for (size_t t = 0; t < c->tensors.size(); ++t) {
            Tensor& tt = c->tensors[t];
            if (tt.last < tt.def) tt.last = tt.def;
            dies[(size_t)tt.last].push_back((int)t);
        }
        size_t top = 0;
        size_t ti = 0;
        for (size_t o = 0; o < c->ops.size(); ++o) {
            // tensors defined by op o are allocated before tensors dying at o are released (an op never aliases in/out)
            for (; ti < c->tensors.size() && c->tensors[ti].def == (int)o; ++ti) {
                Tensor& tt = c->tensors[ti];
                const size_t need = (tt.per_sample() + 63) & ~(size_t)63;
                int best = -1;
                if (!c->debug_taps)
                    for (size_t f = 0; f < freel.size(); ++f)
                        if (freel[f].size >= need && (best < 0 || freel[f].size < freel[(size_t)best].size)) best = (int)f;
                if (best >= 0) {
                    tt.off = freel[(size_t)best].off;
                    if (freel[(size_t)best].size > need) { freel[(size_t)best].off += need; freel[(size_t)best].size -= need; }
                    else freel.erase(freel.begin() + best);
                } else { tt.off = top; top += need; }
            }
            for (int t : dies[o]) {
                const Tensor& tt = c->tensors[(size_t)t];
                freel.push_back({tt.off, (tt.per_sample() + 63) & ~(size_t)63});
            }
        }
        c->ws_per_sample = top;
    }

    // ---- device allocations
    c->w_floats = b.wf;
    HIP_OK(hipMalloc((void**)&c->d_w, c->w_floats * sizeof(float)));
    HIP_OK(hipMemset(c->d_w, 0, c->w_floats * sizeof(float)));
    HIP_OK(hipMalloc((void**)&c->d_int, std::max<size_t>(b.ints.size(), 1) * sizeof(int)));
    HIP_OK(hipMemcpy(c->d_int, b.ints.data(), b.ints.size() * sizeof(int), hipMemcpyHostToDevice));
    const size_t NBmax = (size_t)c->max_batch;
    HIP_OK(hipMalloc((void**)&c->ws, c->ws_per_sample * NBmax * sizeof(float)));
    HIP_OK(hipMalloc((void**)&c->d_jobs, c->jobs.size() * sizeof(PackJob)));
    const size_t Mp = (size_t)pad16(c->max_batch);
    HIP_OK(hipMalloc((void**)&c->d_h1, Mp * T * sizeof(float)));
    HIP_OK(hipMalloc((void**)&c->d_temb, Mp * T * sizeof(float)));
    HIP_OK(hipMalloc((void**)&c->d_dense, Mp * dt * sizeof(float)));
    const size_t E = (size_t)c->H * c->W * a.channels;
    HIP_OK(hipMalloc((void**)&c->d_s2, NBmax * E * sizeof(float)));
    HIP_OK(hipMalloc((void**)&c->d_score, NBmax * E * sizeof(float)));
    HIP_OK(hipMalloc((void**)&c->d_z, NBmax * E * sizeof(float)));
    HIP_OK(hipMalloc((void**)&c->d_norms, (2 * NBmax + 2) * sizeof(float)));
    HIP_OK(hipMalloc((void**)&c->d_tvec, NBmax * sizeof(float)));
    HIP_OK(hipMalloc((void**)&c->d_state, sizeof(StepState)));
    HIP_OK(hipMemset(c->d_state, 0, sizeof(StepState)));
    for (auto& j : c->jobs) j.dst = c->d_w + reinterpret_cast<size_t>(j.dst);

    // ---- resolve pointers that do not depend on parameters
    for (auto& op : c->ops) {
        auto tptr = [&](int t) -> float* { return t >= 0 ? c->ws + c->tensors[(size_t)t].off * NBmax : nullptr; };
        if (op.kind == OP_CONV) {
            ConvArgs& ca = op.conv;
            ca.srcA = tptr(op.tA); ca.srcB = tptr(op.tB); ca.scA = tptr(op.tScA); ca.scB = tptr(op.tScB);
            ca.resid = tptr(op.tRes);
            ca.out = tptr(op.out_tensor);
            ca.tab = c->d_int + op.tab_off;
            ca.mapA = op.has_mapA ? c->d_int + op.mapA_off : nullptr;
            ca.mapSc = op.has_mapSc ? c->d_int + op.mapSc_off : nullptr;
            ca.wpk = c->d_w + op.w_off;
            ca.wsc = ca.Csc ? c->d_w + op.wsc_off : nullptr;
            ca.dense = op.use_dense ? c->d_dense : nullptr;
        } else {
            AttnArgs& aa = op.attn;
            aa.x = tptr(op.tA); aa.out = tptr(op.out_tensor);
            aa.wqkv = c->d_w + op.w_off; aa.bqkv = c->d_w + op.bqkv_off; aa.w3 = c->d_w + op.w3_off;
        }
    }
    return build_fused_program(c);
}

// ------------------------------------------------------------------------------------------
// fused program (workgroup-resident U-Net, csrc/unet_kernel.h)
// ------------------------------------------------------------------------------------------
struct FusedBuilder {
    rdmi_ctx* c;
    static constexpr int LDS_TOTAL = 160 * 1024;
    struct Blk { int off, size; };
    std::vector<Blk> freel;
    int arena_lo = 0, high_water = 0;
    std::map<std::string, int> tabcache;       // geometry key -> encoded table reference (region << 24 | offset in shorts)
    std::vector<short> tabs1;                  // region 1: tables of the multi-sample (low-resolution) section, resident in LDS only during it
    int tab1_lds = 0;                          // LDS byte offset of region 1 (an arena block of the low-resolution section)
    struct LT { int off = -1, C = 0, H = 0, W = 0, rs = 0, bytes = 0, ns = 1; int hw() const { return H * W; } int rows() const { return ns * H * W; } };
    int S = 1;                      // samples per workgroup of the program being built
    bool coop = false;              // co-operative program: S = 4 samples per GROUP of four workgroups (unet_kernel.h: fop_conv_coop)
    bool train = false;             // training forward: stash every layer-plan tensor, Dropout_0 in the fused GroupNorm_1 epilogues
    std::map<std::string, int> lp_tensor, lp_op;      // layer-plan tensor / op indices by name (train)
    int n_xchg = 0;                 // exchanges emitted so far
    int xslot_granules = 0;         // largest exchanged block, in 8-byte granules
    struct PendX { bool on = false; LT t; } pendx;      // output of the last co-operative 3x3 conv: exchanged before the next op is emitted
    int cur_samp = 0;               // sample slot the emitted ops belong to; -1: ops cover all S samples (low-resolution section)
    int multi_slot_off = -1;        // LDS offset of the per-(sample, group) partial-sum slots of multi-sample fused GroupNorms
    int ns_now() const { return cur_samp < 0 ? S : 1; }
    int shift_of(int hw) { int sh = 0; while ((1 << sh) < hw) ++sh; if (cur_samp < 0 && (1 << sh) != hw) fail_("multi-sample ops need a power-of-two pixel count"); return cur_samp < 0 ? sh : 0; }
    size_t spill_floats = 0;
    bool failed = false;
    std::string why;
    int gn_slot_off = 0;            // LDS byte offset of the fused-GroupNorm partial-sum slots (2 KiB)
    bool gn_fuse = true;            // RDMI_NO_GNFUSE=1 keeps every GroupNorm its own op (A/B and debugging)
    int n_gn_fused = 0, n_gn_ops = 0;

    void fail_(const std::string& m) { if (!failed) { failed = true; why = m; } }

    // ---- LDS arena: first-fit with coalescing free list
    void arena_init(int lo) { arena_lo = lo; freel.clear(); freel.push_back({lo, LDS_TOTAL - lo}); high_water = lo; }
    int alloc_bytes(int n) {
        n = (n + 15) & ~15;
        for (size_t i = 0; i < freel.size(); ++i)
            if (freel[i].size >= n) {
                const int o = freel[i].off;
                freel[i].off += n; freel[i].size -= n;
                if (freel[i].size == 0) freel.erase(freel.begin() + (long)i);
                high_water = std::max(high_water, o + n);
                return o;
            }
        {
            std::string fl;
            for (auto& f : freel) fl += " [" + std::to_string(f.off) + "+" + std::to_string(f.size) + "]";
            fail_("LDS arena exhausted at op " + std::to_string(c->fprog.size()) + " (need " + std::to_string(n) + " B; free:" + fl + ")");
        }
        return arena_lo;
    }
    int alloc_top(int n) {                      // odd-sized scratch (Vt, P): carve from the END of the highest block that fits
        n = (n + 15) & ~15;
        for (size_t k = freel.size(); k-- > 0;)
            if (freel[k].size >= n) {
                const int o = freel[k].off + freel[k].size - n;
                freel[k].size -= n;
                if (freel[k].size == 0) freel.erase(freel.begin() + (long)k);
                high_water = std::max(high_water, o + n);
                return o;
            }
        return alloc_bytes(n);
    }
    void free_bytes(int off, int n) {
        n = (n + 15) & ~15;
        freel.push_back({off, n});
        std::sort(freel.begin(), freel.end(), [](const Blk& a, const Blk& b) { return a.off < b.off; });
        for (size_t i = 0; i + 1 < freel.size();)
            if (freel[i].off + freel[i].size == freel[i + 1].off) { freel[i].size += freel[i + 1].size; freel.erase(freel.begin() + (long)i + 1); }
            else ++i;
    }
    LT talloc(int C, int H, int W) {
        LT t; t.C = C; t.H = H; t.W = W; t.ns = ns_now(); t.rs = C + 4; t.bytes = t.ns * H * W * t.rs * 4; t.off = alloc_bytes(t.bytes);
        return t;
    }
    void tfree(LT& t) { if (t.off >= 0) free_bytes(t.off, t.bytes); t.off = -1; }

    // ---- row tables (int16), deduplicated by geometry
    int add_table(const std::string& key, const std::vector<short>& v) {
        auto it = tabcache.find(key);
        if (it != tabcache.end()) return it->second;
        const int region = (S > 1 && cur_samp < 0) ? 1 : 0;
        std::vector<short>& tb = region ? tabs1 : c->ftabs;
        while (tb.size() % 2) tb.push_back(-1);
        const int off = (int)tb.size() | (region << 24);
        tb.insert(tb.end(), v.begin(), v.end());
        tabcache[key] = off;
        return off;
    }
    // tap table of a conv reading tensor (Hs x Ws) seen as a virtual (Hv x Wv) grid (nearest), output (Ho x Wo)
    int tap_table(int Hs, int Ws, int Hv, int Wv, int Ho, int Wo, int stride, int pad_lo, int tap) {
        char key[128];
        const int ns = ns_now();                       // multi-sample: block-diagonal (row s*HWo + p reads source row s*HWs + ...)
        snprintf(key, sizeof key, "tap:%d,%d,%d,%d,%d,%d,%d,%d,%d,%d", Hs, Ws, Hv, Wv, Ho, Wo, stride, pad_lo, tap, ns);
        const int Mpad = pad16(ns * Ho * Wo);
        std::vector<short> v((size_t)Mpad, (short)-1);
        for (int m = 0; m < ns * Ho * Wo; ++m) {
            const int sm = m / (Ho * Wo), pm = m % (Ho * Wo);
            const int oy = pm / Wo, ox = pm % Wo;
            const int iy = oy * stride + tap / 3 - pad_lo, ix = ox * stride + tap % 3 - pad_lo;
            if (iy < 0 || iy >= Hv || ix < 0 || ix >= Wv) continue;
            const int sy = std::min((int)std::floor(iy * ((float)Hs / Hv)), Hs - 1);
            const int sx = std::min((int)std::floor(ix * ((float)Ws / Wv)), Ws - 1);
            v[(size_t)m] = (short)(sm * Hs * Ws + sy * Ws + sx);
        }
        return add_table(key, v);
    }
    int ident_table(int M) {
        const int Mpad = pad16(M);
        std::vector<short> v((size_t)Mpad, (short)-1);
        for (int m = 0; m < M; ++m) v[(size_t)m] = (short)m;
        return add_table("id:" + std::to_string(M), v);
    }
    int map_table(int Hs, int Ws, int Hd, int Wd) {      // nearest map as a GATHER row map
        std::vector<short> v((size_t)Hd * Wd);
        for (int y = 0; y < Hd; ++y)
            for (int x = 0; x < Wd; ++x) {
                const int sy = std::min((int)std::floor(y * ((float)Hs / Hd)), Hs - 1);
                const int sx = std::min((int)std::floor(x * ((float)Ws / Wd)), Ws - 1);
                v[(size_t)y * Wd + x] = (short)(sy * Ws + sx);
            }
        char key[64];
        snprintf(key, sizeof key, "map:%d,%d,%d,%d", Hs, Ws, Hd, Wd);
        return add_table(key, v);
    }

    // ---- op emission.  Table offsets are emitted in SHORTS relative to the table region and turned into LDS
    //      byte offsets once the region's base is known (finish()).
    FOp blank(int kind) {
        FOp o;
        std::memset(&o, 0, sizeof o);
        o.kind = kind; o.a_off = -1; o.b_off = -1; o.a_map_off = -1; o.dense_off = -1; o.resid_off = -1; o.scale = 1.f; o.src_off = -1; o.gn_off = -1; o.samp = cur_samp; o.drop_op = -1;
        return o;
    }
    int emit(const FOp& o) { flush_xchg(); c->fprog.push_back(o); return (int)c->fprog.size() - 1; }
    // all-gather of tensor t (column slices of 32 per member) between the four members; t2: a second tensor of the same shape
    int emit_xchg(const LT& t, const LT* own_rows_src) {
        FOp o = blank(FOP_XCHG);
        o.dst_off = t.off; o.dst_rs = t.rs; o.xidx = n_xchg++;
        if (own_rows_src) {          // row slices: every member contributes its sample's rows (all columns)
            o.a_hw = 1; o.rows = own_rows_src->hw(); o.C = t.C; o.a_off = own_rows_src->off; o.a_rs = own_rows_src->rs;
            if (own_rows_src->C != t.C || t.rows() != 4 * o.rows) fail_("exchange: row-slice shapes");
        } else {
            o.a_hw = 0; o.rows = t.rows(); o.C = t.C / 4;
        }
        if (o.C < 32 || o.C > 128 || (o.C & (o.C - 1)) || o.rows * o.C > 512) fail_("exchange: block of " + std::to_string(o.rows) + " x " + std::to_string(o.C));
        xslot_granules = std::max(xslot_granules, 2 * o.rows * o.C);      // room for a second tensor (<= 1024 granules = 512 pairs: one per thread)
        c->fprog.push_back(o);
        return (int)c->fprog.size() - 1;
    }
    void flush_xchg() { if (pendx.on) { pendx.on = false; emit_xchg(pendx.t, nullptr); } }
    enum { F_GAMMA, F_BETA, F_BIAS, F_BIAS2, F_W, F_SC0W, F_SC1W, F_BIAS_ARENA };
    void patch_param(int op, int field, const std::string& param) { c->fpatch.push_back({op, field, param, 0}); }
    void patch_arena(int op, int field, size_t off) { c->fpatch.push_back({op, field, "", off}); }

    void gather_x(const LT& dst) {               // network input -> LDS [HW][16+4]
        FOp o = blank(FOP_GATHER);
        o.dst_off = dst.off; o.dst_rs = dst.rs; o.rows = dst.rows(); o.C = dst.C;
        o.CA = c->arch.channels; o.CB = 0; o.a_off = -2; o.a_hw = dst.hw(); o.hw_shift = shift_of(dst.hw());
        emit(o);
    }
    // dst = a tensor parked in this workgroup's spill slot `spill` ([n][hw][C]); single- or multi-sample by the current mode
    void gather_g(const LT& dst, size_t spill) {
        FOp o = blank(FOP_GATHER);
        o.dst_off = dst.off; o.dst_rs = dst.rs; o.rows = dst.rows(); o.C = dst.C;
        o.CA = dst.C; o.CB = 0; o.a_off = -1; o.a_hw = dst.hw(); o.a_mod = 0; o.hw_shift = shift_of(dst.hw());
        const int idx = emit(o);
        spill_fix.push_back({idx, spill, 2});
    }
    // dst = concat(A (LDS tensor, nearest-mapped onto dst's grid), B (global spill slot))
    void gather_cat(const LT& dst, const LT& A, size_t spillB, int CB) {
        FOp o = blank(FOP_GATHER);
        o.dst_off = dst.off; o.dst_rs = dst.rs; o.rows = dst.rows(); o.C = dst.C;
        o.CA = A.C; o.CB = CB; o.a_off = A.off; o.a_rs = A.rs; o.a_hw = A.hw(); o.hw_shift = shift_of(dst.hw());
        if (A.H != dst.H || A.W != dst.W) o.a_map_off = map_table(A.H, A.W, dst.H, dst.W);
        o.b_off = -1;
        const int idx = emit(o);
        spill_fix.push_back({idx, spillB, 1});
    }
    void copy_t(const LT& dst, const LT& src) {
        FOp o = blank(FOP_GATHER);
        o.dst_off = dst.off; o.dst_rs = dst.rs; o.rows = dst.rows(); o.C = dst.C;
        o.CA = src.C; o.CB = 0; o.a_off = src.off; o.a_rs = src.rs; o.a_hw = src.hw(); o.hw_shift = shift_of(dst.hw());
        emit(o);
    }
    struct SpillFix { int op; size_t off; int which; };      // which: 0 STORE destination, 1 GATHER source B, 2 GATHER source A
    // Per-sample sections of a multi-sample program are emitted once per sample slot: the spill slots they use must be the
    // SAME for every slot (the buffer is [slot][n][rows][C]); the first emission records them, the others replay them.
    std::vector<size_t> slot_log; size_t slot_pos = 0; bool slot_replay = false;
    std::vector<SpillFix> spill_fix;
    size_t spill_store(const LT& t) {            // LDS tensor -> this workgroup's slot of the spill buffer
        FOp o = blank(FOP_STORE);
        o.dst_off = t.off; o.dst_rs = t.rs; o.rows = t.rows(); o.C = t.C; o.hw_shift = shift_of(t.hw());
        const int idx = emit(o);
        size_t off;
        if (slot_replay) { if (slot_pos >= slot_log.size()) { fail_("spill replay out of slots"); return 0; } off = slot_log[slot_pos++]; }
        else { off = spill_floats; spill_floats += (size_t)t.hw() * t.C; slot_log.push_back(off); }
        spill_fix.push_back({idx, off, 0});
        return off;
    }
    // GroupNorm (+SiLU) of tensor t (in place), or of `src` into t (copy form).  When the tensor being normalised is the
    // output of the CONV op emitted just before (only STOREs in between), has 4-channel groups and the conv's tiling gives
    // every wave a single pass, the GroupNorm is folded into that conv's epilogue (FOp::gn_*) and no op is emitted.
    bool try_fuse_gn(const LT& t, const std::string& pre, bool act, const LT* src) {
        if (!gn_fuse) return false;
        const LT& in = src ? *src : t;
        int j = (int)c->fprog.size() - 1, jx = -1;
        while (j >= 0 && (c->fprog[(size_t)j].kind == FOP_STORE || c->fprog[(size_t)j].kind == FOP_XCHG)) { if (c->fprog[(size_t)j].kind == FOP_XCHG) jx = j; --j; }
        if (j < 0) return false;
        FOp& p = c->fprog[(size_t)j];
        if (p.coop && !pendx.on && (jx < 0 || c->fprog[(size_t)jx].dst_off != in.off || c->fprog[(size_t)jx].src_off >= 0)) return false;
        if (p.kind != FOP_CONV || p.dst_kind != 0 || p.gn_off >= 0 || p.dst_off != in.off || p.dst_rs != in.rs) return false;
        if (p.Cout != t.C || p.rows != t.rows() || t.C % 4 != 0 || std::min(t.C / 4, 32) * 4 != t.C) return false;      // Cg == 4 only
        const int ntiles = p.Cout_pad >> 4;
        const int lWN = (ntiles >= 8 && (ntiles & 7) == 0) ? 3 : (ntiles >= 4 ? 2 : (ntiles >= 2 ? 1 : 0));      // as fop_conv
        const int WN = 1 << lWN, WM = UW_WAVES >> lWN;
        if (ntiles > WN || p.mtiles > 4 * WM) return false;                    // a wave would make several passes
        const int nslots = p.samp >= 0 ? WM * 4 : 4;
        if (p.samp >= 0) { if ((t.C / 4) * nslots * 8 > 1024) return false; }
        else if (multi_slot_off < 0 || t.ns * (t.C / 4) * 32 > 4096 || (t.hw() != 16 && t.hw() != 4)) return false;
        p.gn_off = t.off; p.gn_rs = t.rs; p.gn_act = act ? 1 : 0; p.gn_raw = src ? 1 : 0;
        p.gn_slot_off = p.samp >= 0 ? gn_slot_off : multi_slot_off; p.gn_nslots = nslots;
        p.eps = 1e-6f; p.inv_cnt = 1.0f / (float)(4 * t.hw());
        patch_param(j, F_GAMMA, pre + ".weight"); patch_param(j, F_BETA, pre + ".bias");
        ++n_gn_fused;
        if (train && pre.size() > 12 && pre.compare(pre.size() - 12, 12, ".GroupNorm_1") == 0) {      // this activation is Conv_1's input: Dropout_0 acts on it
            auto it = lp_op.find(pre.substr(0, pre.size() - 12));
            if (it == lp_op.end()) fail_("training forward: no layer-plan op for " + pre);
            else if (c->ops[(size_t)it->second].dropout) p.drop_op = it->second;
        }
        if (p.coop) {        // the conv now also produces the activated tensor (own columns): it has to travel too
            if (pendx.on) {
                if (src) { pendx.on = false; const int ix = emit_xchg(in, nullptr); c->fprog[(size_t)ix].src_off = t.off; c->fprog[(size_t)ix].src_rs = t.rs; }
                // in-place form: the pending exchange of the conv's destination already carries the activated values
            } else if (src) { c->fprog[(size_t)jx].src_off = t.off; c->fprog[(size_t)jx].src_rs = t.rs; }
        }
        return true;
    }
    void gn(const LT& t, const std::string& pre, bool act, const LT* src = nullptr) {
        if (try_fuse_gn(t, pre, act, src)) return;
        if (train && pre.size() > 12 && pre.compare(pre.size() - 12, 12, ".GroupNorm_1") == 0) fail_("training forward: " + pre + " is not folded into its conv (Dropout_0 lives in that epilogue)");
        ++n_gn_ops;
        FOp o = blank(FOP_GN);
        if (src) { o.src_off = src->off; o.src_rs = src->rs; }
        o.dst_off = t.off; o.dst_rs = t.rs; o.rows = t.rows(); o.C = t.C;
        o.G = std::min(t.C / 4, 32); o.act = act ? 1 : 0; o.eps = 1e-6f; o.hw_shift = shift_of(t.hw());
        {
            // multi-sample ops spread the ns * G (sample, group) pairs over the workgroup: T lanes per pair
            const int pairs = t.ns * o.G, T = UW_THREADS / pairs, c4n = t.C / 4;
            o.logT = T == 32 ? 5 : (T == 16 ? 4 : (T == 64 ? 6 : (T == 8 ? 3 : 2)));
            o.logG = 0; while ((1 << o.logG) < o.G) ++o.logG;
            o.Cg = t.C / o.G;
            o.magic_c4n = (65536 + c4n - 1) / c4n; o.magic_Cg = (65536 + o.Cg - 1) / o.Cg;
            o.inv_cnt = 1.0f / (float)(o.Cg * t.hw());
            if ((1 << o.logT) != T || T * pairs != UW_THREADS) fail_("GroupNorm lanes-per-group not a power of two");
            if ((1 << o.logG) != o.G) fail_("GroupNorm group count not a power of two");
            if (t.ns * o.G * 8 > 1024) fail_("GroupNorm statistics of all samples exceed the scratch region");
            if (ceil_div(t.hw(), T) > 6 || o.Cg > 8) fail_("GroupNorm group too large for the register-resident statistics");
        }
        if (t.C % o.G != 0 || UW_THREADS % o.G != 0 || UW_THREADS / o.G > 64) fail_("GroupNorm shape C=" + std::to_string(t.C));
        const int idx = emit(o);
        patch_param(idx, F_GAMMA, pre + ".weight"); patch_param(idx, F_BETA, pre + ".bias");
    }
    struct ConvOut { int kind; LT* t; };
    // 3x3 conv (or 1x1 when ntap == 1) over LDS tensor `in`
    int conv(const LT& in, int Hv, int Wv, int Ho, int Wo, int stride, int pad_lo, int ntap, const std::string& wkey,
             size_t w_extra_off, int Cout, const std::string& bias_param, size_t bias_arena, int dst_kind, const LT* dst,
             float scale, int dense_off, const LT* resid, const LT* sc_src, const std::string& sc_key,
             const std::string& bias2_param) {
        FOp o = blank(FOP_CONV);
        const int ns = ns_now();
        o.rows = ns * Ho * Wo; o.mtiles = pad16(o.rows) / 16; o.Cout = Cout; o.Cout_pad = pad16(Cout);
        o.ntap = ntap; o.hw_shift = shift_of(Ho * Wo);
        if (ns > 1 && dst_kind != 0) fail_("multi-sample convs write LDS tensors only");
        o.main_ph.lds_off = in.off; o.main_ph.rs = in.rs; o.main_ph.nch = pad16(in.C) / 16;
        if (in.C % 16 != 0) fail_("conv input channels not padded");
        for (int t = 0; t < ntap; ++t)
            o.tab_off[t] = ntap == 1 ? ident_table(ns * Ho * Wo) : tap_table(in.H, in.W, Hv, Wv, Ho, Wo, stride, pad_lo, t);
        o.dense_off = dense_off; o.scale = scale; o.dst_kind = dst_kind;
        if (dst) { o.dst_off = dst->off; o.dst_rs = dst->rs; }
        if (resid) { o.resid_off = resid->off; o.resid_rs = resid->rs; }
        if (o.mtiles > 6) fail_("conv with more than 96 output rows per workgroup");
        if (sc_src) fail_("fused CONV ops carry no shortcut phases (the NIN shortcut is its own 1x1 op)");
        if (sc_src) {
            o.nsc = 1;
            o.sc[0].lds_off = sc_src->off; o.sc[0].rs = sc_src->rs; o.sc[0].nch = sc_src->C / 16; o.sc[0].tab_off = ident_table(ns * Ho * Wo);
        }
        const int idx = emit(o);
        patch_arena(idx, F_W, c->wmap.at(wkey) + w_extra_off);
        if (!bias_param.empty()) patch_param(idx, F_BIAS, bias_param); else patch_arena(idx, F_BIAS_ARENA, bias_arena);
        if (sc_src) { patch_arena(idx, F_SC0W, c->wmap.at(sc_key)); patch_param(idx, F_BIAS2, bias2_param); }
        if (train) {                     // the layer plan keeps this conv's output under the name of its op: "<block>.Conv_0", "<block>" (Conv_1), the attention block, or the conv itself
            std::string tn = wkey;
            auto ends = [&](const char* suf) { const size_t n = std::strlen(suf); return tn.size() >= n && tn.compare(tn.size() - n, n, suf) == 0; };
            if (ends(".Conv_1")) tn.resize(tn.size() - 7);
            else if (ends(".NIN_3")) tn.resize(tn.size() - 6);
            else if (ends(".NIN_0") || ends(".qkv3")) tn.clear();
            else if (ends(".Conv_0") && (tn.rfind("downsample.", 0) == 0 || tn.rfind("upsample.", 0) == 0)) tn.resize(tn.size() - 7);
            if (!tn.empty() && dst_kind == 0) {
                auto it = lp_tensor.find(tn);
                if (it == lp_tensor.end()) fail_("training forward: no layer-plan tensor '" + tn + "'");
                else {
                    const Tensor& t = c->tensors[(size_t)it->second];
                    FOp& q = c->fprog[(size_t)idx];
                    if (t.C != Cout || t.H * t.W != q.rows) fail_("training forward: tensor '" + tn + "' has another shape");
                    q.stash = c->ws + t.off * (size_t)c->max_batch;
                    q.stash_bf16 = c->arch.compute_dtype == 1 ? 1 : 0;
                }
            }
            c->fprog[(size_t)idx].drop_op = -1;
        }
        if (coop && cur_samp < 0) {      // low-resolution section of the co-operative program: column-sliced, K split over the waves
            FOp& q = c->fprog[(size_t)idx];
            q.coop = 1;
            if (q.Cout_pad != 128 || q.mtiles != 1 || q.main_ph.nch % 4 != 0 || dst_kind != 0 || !dst || dst->C != 128)
                fail_("co-operative conv shape (Cout " + std::to_string(Cout) + ", " + std::to_string(q.mtiles) + " row tiles, " + std::to_string(q.main_ph.nch) + " chunks)");
            if (ntap == 9) { pendx.on = true; pendx.t = *dst; }      // the next conv contracts over this tensor: all-gather it
        }
        return idx;
    }
};

// ResnetBlockDDPMpp on LDS tensors.  xin: raw block input (consumed).  Returns the block output.
// When the block has a NIN shortcut (cin != cout, i.e. every concat-fed up block) the shortcut is evaluated FIRST
// into the output buffer, so GroupNorm_0 can then run in place on xin: no second copy of the (large) concat
// tensor is ever resident.  Identity-shortcut blocks keep xin raw for the residual and normalise a copy.
FusedBuilder::LT fused_resblock(FusedBuilder& b, const std::string& name, FusedBuilder::LT& xin, int cout, int dense_off) {
    using LT = FusedBuilder::LT;
    const int H = xin.H, W = xin.W;
    const float rs2 = (float)(1.0 / std::sqrt(2.0));
    if (xin.C != cout) {
        LT out = b.talloc(cout, H, W);
        b.conv(xin, H, W, H, W, 1, 0, 1, name + ".NIN_0", 0, cout, name + ".NIN_0.b", 0, 0, &out, 1.f, -1, nullptr, nullptr, "", "");
        b.gn(xin, name + ".GroupNorm_0", true);
        LT h1 = b.talloc(cout, H, W);
        b.conv(xin, H, W, H, W, 1, 1, 9, name + ".Conv_0", 0, cout, name + ".Conv_0.bias", 0, 0, &h1, 1.f, dense_off, nullptr, nullptr, "", "");
        b.tfree(xin);
        b.gn(h1, name + ".GroupNorm_1", true);
        b.conv(h1, H, W, H, W, 1, 1, 9, name + ".Conv_1", 0, cout, name + ".Conv_1.bias", 0, 0, &out, rs2, -1, &out, nullptr, "", "");
        b.tfree(h1);
        return out;
    }
    LT xact = b.talloc(xin.C, H, W);
    b.gn(xact, name + ".GroupNorm_0", true, &xin);
    LT h1 = b.talloc(cout, H, W);
    b.conv(xact, H, W, H, W, 1, 1, 9, name + ".Conv_0", 0, cout, name + ".Conv_0.bias", 0, 0, &h1, 1.f, dense_off, nullptr, nullptr, "", "");
    b.tfree(xact);
    b.gn(h1, name + ".GroupNorm_1", true);
    LT out = b.talloc(cout, H, W);
    b.conv(h1, H, W, H, W, 1, 1, 9, name + ".Conv_1", 0, cout, name + ".Conv_1.bias", 0, 0, &out, rs2, -1, &xin, nullptr, "", "");
    b.tfree(h1);
    b.tfree(xin);
    return out;
}

// AttnBlockpp on an LDS tensor (consumed).  Returns the block output.
FusedBuilder::LT fused_attn(FusedBuilder& b, const std::string& name, FusedBuilder::LT& x) {
    using LT = FusedBuilder::LT;
    rdmi_ctx* c = b.c;
    const int C = x.C, H = x.H, W = x.W, L = H * W, Lpad = pad16(L);
    if (C != 64 || Lpad > 96) { b.fail_("attention: only C=64, H*W<=96 is built"); return x; }
    if (b.cur_samp < 0) { b.fail_("attention inside the multi-sample (low-resolution) section is not built"); return x; }
    LT xn = b.talloc(C, H, W);
    b.gn(xn, name + ".GroupNorm_0", false, &x);
    LT q = b.talloc(C, H, W), k = b.talloc(C, H, W);
    const int ps = Lpad + 4;
    const int vt_bytes = C * ps * 4, p_bytes = L * ps * 4;
    const int vt_off = b.alloc_top(vt_bytes);
    // Inference programs fold NIN_3 into the value projection (weights packed as ".qkv3f" / ".bqkvf"): the attention op then writes
    // (P V' + b3 + x) / sqrt2 itself and the 1x1 NIN_3 conv -- 10-15 k cycles of almost pure per-op overhead, five times per forward --
    // disappears.  The training forward keeps the reference's op sequence (the backward needs the tensor between P V and NIN_3).
    const bool fold3 = !b.train && std::getenv("RDMI_NO_ATTN_FOLD") == nullptr;
    const size_t bq = c->wmap.at(name + (fold3 ? ".bqkvf" : ".bqkv"));
    {   // q, k, v in one contraction: the layer plan already packs NIN_0..2 side by side? no: [3][C/16][C][16] -> use the
        // dedicated fused packing [C/16][3C][16] (wmap key ".qkv3")
        const int idx = b.conv(xn, H, W, H, W, 1, 0, 1, name + (fold3 ? ".qkv3f" : ".qkv3"), 0, 3 * C, "", bq, 3, &q, 1.f, -1, nullptr, nullptr, "", "");
        FOp& o = c->fprog[(size_t)idx];
        o.dst2_off = k.off; o.dst3_off = vt_off; o.dst3_rs = ps; o.split_C = C;
        o.qkv1 = (std::getenv("RDMI_NO_QKV1") == nullptr && o.ntap == 1 && o.main_ph.nch == 4 && o.Cout_pad == 192 && C == 64 && o.mtiles <= 6) ? 1 : 0;
    }
    b.tfree(xn);
    const int p_off = b.alloc_top(p_bytes);
    LT o = b.talloc(C, H, W);
    {
        FOp a = b.blank(FOP_ATTN);
        a.q_off = q.off; a.k_off = k.off; a.qk_rs = q.rs; a.vt_off = vt_off; a.p_off = p_off; a.ps = ps;
        a.L = L; a.Lpad = Lpad; a.C = C; a.att_scale = 1.0f / std::sqrt((float)C);
        a.dst_off = o.off; a.dst_rs = o.rs;
        if (fold3) { a.resid_off = x.off; a.resid_rs = x.rs; a.scale = (float)(1.0 / std::sqrt(2.0)); }
        const int ia = b.emit(a);
        if (fold3) b.patch_param(ia, FusedBuilder::F_BIAS, name + ".NIN_3.b");
    }
    b.tfree(q); b.tfree(k);
    b.free_bytes(vt_off, vt_bytes); b.free_bytes(p_off, p_bytes);
    if (fold3) { b.tfree(x); return o; }
    LT out = b.talloc(C, H, W);
    b.conv(o, H, W, H, W, 1, 0, 1, name + ".NIN_3", 0, C, name + ".NIN_3.b", 0, 0, &out, (float)(1.0 / std::sqrt(2.0)), -1, &x, nullptr, "", "");
    b.tfree(o);
    b.tfree(x);
    return out;
}

int build_fused_program_pass(rdmi_ctx* c, int S, int TAB_RESERVE, int TAB1_RESERVE, int* tab_used, int* tab1_used, bool coop, bool train);

// A built program is moved out of the context's construction fields into one of these (one per samples-per-workgroup value).
void stash_program(rdmi_ctx* c, int S) {
    rdmi_ctx::FusedProg q;
    q.S = S; q.ok = c->fused_ok; q.why = c->fused_why;
    q.fprog.swap(c->fprog); q.fpatch.swap(c->fpatch); q.ftabs.swap(c->ftabs); q.fdesc.swap(c->fdesc);
    q.d_fprog = c->d_fprog; q.d_ftabs = c->d_ftabs; q.d_spill = c->d_spill; q.spill_per_sample = c->spill_per_sample;
    q.fargs = c->fargs; q.fused_lds = c->fused_lds;
    q.train = c->cur_train != 0; c->cur_train = 0;
    q.coop = c->cur_coop != 0; q.n_xchg = c->cur_nxchg; q.d_xbuf = c->cur_xbuf; q.d_coop_err = c->cur_coop_err; q.max_nb = c->cur_cap_n;
    c->cur_coop = 0; c->cur_nxchg = 0; c->cur_xbuf = nullptr; c->cur_coop_err = nullptr; c->cur_cap_n = 0;
    c->d_fprog = nullptr; c->d_ftabs = nullptr; c->d_spill = nullptr; c->spill_per_sample = 0; c->fused_ok = false; c->fused_why.clear();
    c->progs.push_back(std::move(q));
}

int build_one_program(rdmi_ctx* c, int S, bool coop = false, bool train = false) {
    int used = 0, used1 = 0;
    if (int e = build_fused_program_pass(c, S, 8 * 1024, 24 * 1024, &used, &used1, coop, train)) return e;
    if (c->fused_ok) {   // second pass with the table region sized exactly
        for (void* p : {(void*)c->d_fprog, (void*)c->d_ftabs, (void*)c->d_spill, (void*)c->cur_xbuf, (void*)c->cur_coop_err}) if (p) (void)hipFree(p);
        c->d_fprog = nullptr; c->d_ftabs = nullptr; c->d_spill = nullptr; c->cur_xbuf = nullptr; c->cur_coop_err = nullptr;
    }
    c->fused_ok = false; c->fprog.clear(); c->fpatch.clear(); c->ftabs.clear(); c->fused_why.clear();
    if (int e = build_fused_program_pass(c, S, (used + 63) & ~63, (used1 + 63) & ~63, &used, &used1, coop, train)) return e;
    c->cur_train = train ? 1 : 0;
    stash_program(c, S);
    return 0;
}

// How many workgroups of the fused kernel the device holds at once (one per CU: each takes the CU's whole LDS).
int resident_workgroups() {
#ifdef RDMI_EMU
    return 256;
#else
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    return cus;
#endif
}

int build_fused_program(rdmi_ctx* c) {
    // S = 1: one sample per workgroup (B <= 128 with guidance fills the 256 CUs exactly).  Larger batches run the
    // low-resolution half of the network for S = 2 / 4 samples at once per workgroup: every streamed weight fragment then
    // feeds S samples (that half is weight-stream bound at S = 1) and its per-op fixed cost is shared.
    if (int e = build_one_program(c, 1)) return e;
    const char* only = std::getenv("RDMI_S");           // RDMI_S=1 keeps the single-sample program only (A/B)
    if (const char* e = std::getenv("RDMI_S_MIN_WG")) c->s_min_wg = std::max(1, atoi(e));
    for (int S : {2, 4})
        if (c->progs[0].ok && c->arch.n_levels >= 2 && c->max_batch >= c->s_min_wg * S && (!only || atoi(only) >= S))
            if (int e = build_one_program(c, S)) return e;
    // The co-operative program (groups of four workgroups share the low-resolution section): for batches that fit the chip in one
    // wave of workgroups.  RDMI_COOP=0 leaves it out; RDMI_COOP_STRIDE sets the distance between a group's workgroup ids.
    if (const char* e = std::getenv("RDMI_COOP")) c->use_coop = atoi(e) != 0;
#ifdef RDMI_EMU
    c->coop_stride = 1;          // the emulator runs consecutive workgroup ids concurrently
#endif
    if (const char* e = std::getenv("RDMI_COOP_STRIDE")) c->coop_stride = std::max(1, atoi(e));
    if (c->progs[0].ok && c->use_coop && c->arch.n_levels >= 2 && !only)
        if (int e = build_one_program(c, 4, true)) return e;
    return 0;
}

int build_fused_program_pass(rdmi_ctx* c, int S, int TAB_RESERVE, int TAB1_RESERVE, int* tab_used, int* tab1_used, bool coop, bool train) {
    using LT = FusedBuilder::LT;
    const rdmi_arch& a = c->arch;
    FusedBuilder b{c};
    b.S = S; b.coop = coop; b.train = train;
    if (train) {
        for (size_t i = 0; i < c->tensors.size(); ++i) b.lp_tensor[c->tensors[i].name] = (int)i;
        for (size_t i = 0; i < c->ops.size(); ++i) b.lp_op[c->ops[i].name] = (int)i;
    }
    const int nslot0 = coop ? 1 : S;                  // how often the full-resolution sections are emitted (co-operative: the member's own sample only)
    Layout L = build_layout(c);
    std::map<std::string, int> dense_off;
    {
        int dt = 0;
        for (auto& d : L.down) { dense_off[d.name] = dt; dt += d.cout; }
        dense_off["mid_block1"] = dt; dt += L.mid_ch;
        dense_off["mid_block2"] = dt; dt += L.mid_ch;
        for (auto& u : L.up) { dense_off[u.name] = dt; dt += u.cout; }
    }
    // LDS: [row tables] [zero row] [GN stats] [tensor arena]
    const int zero_bytes = 1280;                     // >= (Cmax/16)*64 + 64 for Cmax = 256... host-checked below
    const int stat_bytes = 1024 + 64;                         // GN op scratch [S][2 x 32] overlaid with the fused-GroupNorm partial-sum slots (never live together)
    b.arena_init(TAB_RESERVE + zero_bytes + stat_bytes);
    b.gn_slot_off = TAB_RESERVE + zero_bytes;
    b.gn_fuse = std::getenv("RDMI_NO_GNFUSE") == nullptr;
    const int H0 = c->H, W0 = c->W;
    if (H0 * W0 > 96) { c->fused_why = "more than 96 pixels per sample"; return 0; }
    const int nlev = a.n_levels;
    const bool multi = S > 1;

    struct HS { size_t spill; int C, H, W; };
    std::vector<HS> hs;                               // the skip stack (RD/models/ncsnpp.py:268-292, 311-338)
    int ch = a.nf, H = H0, W = W0;
    size_t hand_down = 0, hand_up = 0;                // multi-sample: level-0 <-> low-resolution hand-off slots
    LT h;
    // ---- level 0, down path: once per sample slot (same LDS, same spill slots)
    const int nrb = a.num_res_blocks;
    for (int sl = 0; sl < nslot0; ++sl) {
        b.cur_samp = sl;
        b.slot_replay = sl > 0; b.slot_pos = 0;
        H = H0; W = W0; ch = a.nf;
        LT xin = b.talloc(16, H, W);
        b.gather_x(xin);
        h = b.talloc(a.nf, H, W);
        b.conv(xin, H, W, H, W, 1, 1, 9, "input_conv", 0, a.nf, "input_conv.bias", 0, 0, &h, 1.f, -1, nullptr, nullptr, "", "");
        b.tfree(xin);
        if (sl == 0) hs.push_back({(size_t)-1, a.nf, H, W});          // hs[0] (input_conv output) is never popped (RD/models/ncsnpp.py:268,315)
        for (int j = 0; j < nrb; ++j) {
            const BlockSpec& bs = L.down[(size_t)j];
            h = fused_resblock(b, bs.name, h, bs.cout, dense_off[bs.name]);
            ch = bs.cout;
            if (bs.attn) h = fused_attn(b, "down_attn." + std::to_string(j), h);
            const size_t sp = b.spill_store(h);
            if (sl == 0) hs.push_back({sp, ch, H, W});
        }
        if (sl == 0) hs.push_back(hs.back());
        if (nlev > 1) {
            const int Ho = (H + 1 - 3) / 2 + 1, Wo = (W + 1 - 3) / 2 + 1;
            LT o = b.talloc(ch, Ho, Wo);
            const std::string nm = "downsample.0.Conv_0";
            b.conv(h, H, W, Ho, Wo, 2, 0, 9, nm, 0, ch, nm + ".bias", 0, 0, &o, 1.f, -1, nullptr, nullptr, "", "");
            b.tfree(h);
            h = o; H = Ho; W = Wo;
            if (multi && !coop) { hand_down = b.spill_store(h); b.tfree(h); }
        }
    }
    b.slot_replay = false;
    // ---- levels >= 1 (down), bottleneck, levels >= 1 (up): all S samples at once when S > 1.
    //      Co-operative program: only the levels of at most 4 pixels per sample (and the bottleneck) run in the shared form -- that
    //      part is bound by the weight stream and by 4-row MFMA tiles with one sample per CU; the 4x4 level is matrix-pipe bound
    //      either way (measured: sharing it costs more in exchanges than it saves) and stays per sample like level 0.
    int d = nrb, u = 0, loadtab_idx = -1;
    {
        b.cur_samp = (multi && !coop) ? -1 : 0;
        int loadtab_op = -1;
        bool in_coop = false;
        auto enter_multi = [&]() {
            b.multi_slot_off = b.alloc_top(4096);
            b.tab1_lds = b.alloc_top(TAB1_RESERVE);
            FOp lt = b.blank(FOP_LOADTAB);            // the section's row tables: global -> LDS (source / size patched below)
            lt.dst_off = b.tab1_lds;
            loadtab_op = b.emit(lt);
        };
        auto leave_multi = [&]() {
            b.free_bytes(b.multi_slot_off, 4096);
            b.multi_slot_off = -1;
            b.free_bytes(b.tab1_lds, TAB1_RESERVE);
            while (b.tabs1.size() % 8) b.tabs1.push_back(-1);
            c->fprog[(size_t)loadtab_op].rows = (int)b.tabs1.size() * 2;      // bytes
        };
        auto enter_coop = [&]() {                     // h: this member's own sample -> the four samples of the group (rows of sample m come from member m)
            b.flush_xchg();
            b.cur_samp = -1;
            enter_multi();
            LT h4 = b.talloc(ch, H, W);
            b.emit_xchg(h4, &h);
            b.tfree(h);
            h = h4; in_coop = true;
        };
        auto leave_coop = [&]() {                     // back to the member's own sample: its rows of the complete four-sample tensor
            b.flush_xchg();
            b.cur_samp = 0;
            LT own = b.talloc(ch, H, W);
            b.copy_t(own, h);
            c->fprog.back().a_hw = own.hw(); c->fprog.back().a_mstride = own.hw() * h.rs * 4;
            b.tfree(h);
            h = own; in_coop = false;
            leave_multi();
        };
        if (multi && !coop) {
            enter_multi();
            h = b.talloc(ch, H, W);
            b.gather_g(h, hand_down);
        }
        int coop_level = -1;
        for (int i = 1; i < nlev; ++i) {
            if (coop && !in_coop && H * W == 4) { enter_coop(); coop_level = i; }       // 4 pixels per sample: 4 samples = one 16-row MFMA tile
            for (int j = 0; j < nrb; ++j, ++d) {
                const BlockSpec& bs = L.down[(size_t)d];
                h = fused_resblock(b, bs.name, h, bs.cout, dense_off[bs.name]);
                ch = bs.cout;
                if (bs.attn) h = fused_attn(b, "down_attn." + std::to_string(d), h);
                hs.push_back({b.spill_store(h), ch, H, W});
            }
            hs.push_back(hs.back());
            if (i != nlev - 1) {
                const int Ho = (H + 1 - 3) / 2 + 1, Wo = (W + 1 - 3) / 2 + 1;
                LT o = b.talloc(ch, Ho, Wo);
                const std::string nm = "downsample." + std::to_string(i) + ".Conv_0";
                b.conv(h, H, W, Ho, Wo, 2, 0, 9, nm, 0, ch, nm + ".bias", 0, 0, &o, 1.f, -1, nullptr, nullptr, "", "");
                b.tfree(h);
                h = o; H = Ho; W = Wo;
            }
        }
        if (coop && !in_coop) b.fail_("no level of exactly 4 pixels per sample: nothing to share");
        h = fused_resblock(b, "mid_block1", h, ch, dense_off["mid_block1"]);
        h = fused_resblock(b, "mid_block2", h, ch, dense_off["mid_block2"]);
        for (int k = 0; k < nlev - 1; ++k) {          // up levels nlev-1 .. 1
            for (int j = 0; j < nrb + 1; ++j, ++u) {
                const BlockSpec& bs = L.up[(size_t)u];
                HS sk = hs.back();
                hs.pop_back();
                LT cat = b.talloc(ch + sk.C, sk.H, sk.W);
                b.gather_cat(cat, h, sk.spill, sk.C);
                b.tfree(h);
                H = sk.H; W = sk.W;
                h = fused_resblock(b, bs.name, cat, bs.cout, dense_off[bs.name]);
                ch = bs.cout;
                if (bs.attn) h = fused_attn(b, "up_attn." + std::to_string(u), h);
            }
            if (in_coop && nlev - 1 - k == coop_level) leave_coop();      // the levels above run per sample again
            if (k != nlev - 2) {                       // upsample between two low levels stays in this section
                LT o = b.talloc(ch, 2 * H, 2 * W);
                const std::string nm = "upsample." + std::to_string(k) + ".Conv_0";
                b.conv(h, 2 * H, 2 * W, 2 * H, 2 * W, 1, 1, 9, nm, 0, ch, nm + ".bias", 0, 0, &o, 1.f, -1, nullptr, nullptr, "", "");
                b.tfree(h);
                h = o; H *= 2; W *= 2;
            }
        }
        if (in_coop) leave_coop();
        if (multi && !coop) {
            hand_up = b.spill_store(h); b.tfree(h);
            leave_multi();
        }
        loadtab_idx = loadtab_op;
    }
    // ---- level 0, up path: once per sample slot
    const int u0 = u, ch_low = ch, H_low = H, W_low = W;
    const std::vector<HS> hs0 = hs;
    for (int sl = 0; sl < nslot0; ++sl) {
        b.cur_samp = sl;
        u = u0; ch = ch_low; H = H_low; W = W_low; hs = hs0;
        if (multi && !coop) { h = b.talloc(ch, H, W); b.gather_g(h, hand_up); }
        if (nlev > 1) {                                // the upsample conv onto level 0's (even) grid
            LT o = b.talloc(ch, 2 * H, 2 * W);
            const std::string nm = "upsample." + std::to_string(nlev - 2) + ".Conv_0";
            b.conv(h, 2 * H, 2 * W, 2 * H, 2 * W, 1, 1, 9, nm, 0, ch, nm + ".bias", 0, 0, &o, 1.f, -1, nullptr, nullptr, "", "");
            b.tfree(h);
            h = o; H *= 2; W *= 2;
        }
        for (int j = 0; j < nrb + 1; ++j, ++u) {
            const BlockSpec& bs = L.up[(size_t)u];
            HS sk = hs.back();
            hs.pop_back();
            LT cat = b.talloc(ch + sk.C, sk.H, sk.W);
            b.gather_cat(cat, h, sk.spill, sk.C);
            b.tfree(h);
            H = sk.H; W = sk.W;
            h = fused_resblock(b, bs.name, cat, bs.cout, dense_off[bs.name]);
            ch = bs.cout;
            if (bs.attn) h = fused_attn(b, "up_attn." + std::to_string(u), h);
        }
        b.gn(h, "out_norm", true);
        b.conv(h, H, W, H, W, 1, 1, 9, "out_conv", 0, a.channels, "out_conv.bias", 0, 2, nullptr, 1.f, -1, nullptr, nullptr, "", "");
        b.tfree(h);
    }
    if (H != c->H || W != c->W) b.fail_("network output grid differs from the input grid");

    *tab_used = (int)c->ftabs.size() * 2 + 16;
    *tab1_used = (int)b.tabs1.size() * 2 + 16;
    if (*tab_used > TAB_RESERVE && TAB_RESERVE != 8 * 1024) b.fail_("row tables exceed the reserved LDS region");
    if (*tab1_used > TAB1_RESERVE && multi) b.fail_("low-resolution row tables exceed their LDS block");
    int cmax = 16;
    for (auto& o : c->fprog) if (o.kind == FOP_CONV) { cmax = std::max(cmax, o.main_ph.nch * 16); for (int s2 = 0; s2 < o.nsc; ++s2) cmax = std::max(cmax, o.sc[s2].nch * 16); }
    if (cmax * 4 + 64 > zero_bytes) b.fail_("zero row too small");
    if (b.failed && TAB_RESERVE == 8 * 1024 && *tab_used <= 8 * 1024) { c->fprog.clear(); c->fpatch.clear(); c->ftabs.clear(); c->fused_ok = false; return 0; }   // pass 1 only sizes the tables
    if (b.failed) { c->fused_why = b.why; c->fprog.clear(); c->fpatch.clear(); c->ftabs.clear(); return 0; }

    // table offsets (shorts, relative) -> LDS byte offsets
    auto tab_lds = [&](int ref) { return (ref >> 24) ? b.tab1_lds + (ref & 0xffffff) * 2 : (ref & 0xffffff) * 2; };
    for (auto& o : c->fprog) {
        if (o.kind == FOP_CONV) {
            for (int t = 0; t < o.ntap; ++t) o.tab_off[t] = tab_lds(o.tab_off[t]);
            for (int s2 = 0; s2 < o.nsc; ++s2) o.sc[s2].tab_off = tab_lds(o.sc[s2].tab_off);
        } else if (o.kind == FOP_GATHER && o.a_map_off >= 0) {
            o.a_map_off = tab_lds(o.a_map_off);
        }
    }
    while (c->ftabs.size() % 8) c->ftabs.push_back(-1);
    const size_t tab0_shorts = c->ftabs.size();                  // region 0 is what the kernel copies at start; region 1 follows it in the buffer
    c->ftabs.insert(c->ftabs.end(), b.tabs1.begin(), b.tabs1.end());
    c->spill_per_sample = b.spill_floats;
    // spill slots are [slot][n][rows][C].  Co-operative program: the low-resolution slots are indexed by (workgroup, sample of its group),
    // n = 4 * workgroup id + slot, so the buffer holds 4 * (largest grid) "samples" per slot
    int coop_max_nb = 0, spill_n = c->max_batch;
    if (coop) {
        const int st = c->coop_stride, res = resident_workgroups();
        const int max_groups = (res / (4 * st)) * st;                      // whole blocks of 4 * stride workgroup ids that are resident together
        coop_max_nb = std::min(c->max_batch, 4 * max_groups);
        if (coop_max_nb < 1) b.fail_("no room for one co-operative group");
        const int grid_max = ceil_div(ceil_div(std::max(coop_max_nb, 1), 4), st) * 4 * st;
        spill_n = 4 * grid_max;
    }
    if (b.failed) { c->fused_why = b.why; c->fprog.clear(); c->fpatch.clear(); c->ftabs.clear(); return 0; }
    HIP_OK(hipMalloc((void**)&c->d_spill, std::max<size_t>(c->spill_per_sample, 1) * (size_t)spill_n * sizeof(float)));
    HIP_OK(hipMalloc((void**)&c->d_fprog, c->fprog.size() * sizeof(FOp)));
    HIP_OK(hipMalloc((void**)&c->d_ftabs, c->ftabs.size() * sizeof(short)));
    HIP_OK(hipMemcpy(c->d_ftabs, c->ftabs.data(), c->ftabs.size() * sizeof(short), hipMemcpyHostToDevice));
    // spill slots: [slot][n][rows][C] so a workgroup only ever touches its own sample's rows
    for (auto& f : b.spill_fix) {
        FOp& o = c->fprog[(size_t)f.op];
        float* p = c->d_spill + f.off * (size_t)spill_n;
        if (f.which == 1) o.b_g = p; else if (f.which == 2) o.a_g = p; else o.g_out = p;
    }
    c->fargs = UnetArgs{};
    c->fargs.prog = c->d_fprog; c->fargs.nops = (int)c->fprog.size();
    c->fargs.tabs = c->d_ftabs; c->fargs.tab_bytes = (int)(tab0_shorts * sizeof(short)); c->fargs.tab_base = 0;
    if (loadtab_idx >= 0) c->fprog[(size_t)loadtab_idx].a_g = reinterpret_cast<const float*>(c->d_ftabs + tab0_shorts);
    c->fargs.zero_off = TAB_RESERVE; c->fargs.zero_bytes = zero_bytes;
    c->fargs.dense = c->d_dense; c->fargs.dense_stride = c->dense_total; c->fargs.S = S;
    c->fargs.out_elems = c->H * c->W * a.channels;
    if (coop) {
        const int groups_max = ceil_div(coop_max_nb, 4);
        const size_t xb = (size_t)groups_max * 2 * 4 * (size_t)b.xslot_granules * sizeof(unsigned long long);
        HIP_OK(hipMalloc((void**)&c->cur_xbuf, std::max<size_t>(xb, 16)));
        HIP_OK(hipMemset(c->cur_xbuf, 0, std::max<size_t>(xb, 16)));          // tags start at 0; epochs never are
        HIP_OK(hipMalloc((void**)&c->cur_coop_err, 16));
        HIP_OK(hipMemset(c->cur_coop_err, 0, 16));
        c->fargs.coop = 1; c->fargs.coop_stride = c->coop_stride; c->fargs.xbuf = c->cur_xbuf; c->fargs.xslot = b.xslot_granules;
        c->fargs.coop_err = c->cur_coop_err;
        if (const char* e = std::getenv("RDMI_COOP_TEST_BREAK")) c->fargs.coop_break = atoi(e);
        c->cur_coop = 1; c->cur_nxchg = b.n_xchg; c->cur_cap_n = coop_max_nb;
    }
    c->fused_lds = (size_t)b.high_water;
    if (const char* e = std::getenv("RDMI_UDBG")) c->fargs.dbg = atoi(e);
    if (std::getenv("RDMI_STAMPS") && (S == 1 || coop)) {       // diagnostic builds exist for the single-sample and the co-operative program
        if (!c->d_stamps) HIP_OK(hipMalloc((void**)&c->d_stamps, (1024 + 200 * 8 + 8) * sizeof(long long)));
        if (c->fprog.size() > 200) return fail("RDMI_STAMPS: program of %zu ops", c->fprog.size());
        c->fargs.stamps = c->d_stamps;
    }
    c->fdesc.clear();
    for (auto& o : c->fprog) {
        char buf[160];
        const char* kn[] = {"GATHER", "STORE", "GN", "CONV", "ATTN", "LOADTAB", "XCHG"};
        if (o.kind == FOP_CONV) snprintf(buf, sizeof buf, "CONV rows=%d mtiles=%d K=%dx%d(+%d) Cout=%d dst=%d%s", o.rows, o.mtiles, o.ntap, o.main_ph.nch * 16, o.nsc ? o.sc[0].nch * 16 : 0, o.Cout, o.dst_kind, o.gn_off >= 0 ? (o.gn_raw ? (o.coop ? " +GN(copy) coop" : " +GN(copy)") : (o.coop ? " +GN coop" : " +GN")) : (o.coop ? " coop" : ""));
        else snprintf(buf, sizeof buf, "%s rows=%d C=%d", kn[o.kind], o.rows, o.C);
        c->fdesc.push_back(buf);
    }
    c->fused_ok = true;
    return 0;
}

// ------------------------------------------------------------------------------------------
// launches
// ------------------------------------------------------------------------------------------
struct ProfScope {
    rdmi_ctx* c; hipStream_t s; int entry = -1; size_t slot = 0;
    ProfScope(rdmi_ctx* c_, hipStream_t s_, const std::string& name, double flops) : c(c_), s(s_) {
        if (!c->profiling) return;
        for (size_t i = 0; i < c->prof.size(); ++i) if (c->prof[i].name == name) entry = (int)i;
        if (entry < 0) { c->prof.push_back({name, 0, 0, 0}); entry = (int)c->prof.size() - 1; }
        c->prof[(size_t)entry].flops += flops;
        if (c->ev_used == c->ev_pool.size()) {
            hipEvent_t a, b;
            hipEventCreate(&a); hipEventCreate(&b);
            c->ev_pool.push_back({a, b});
            c->ev_entry.push_back(0);
        }
        slot = c->ev_used++;
        c->ev_entry[slot] = entry;
        hipEventRecord(c->ev_pool[slot].first, s);
    }
    ~ProfScope() { if (entry >= 0) hipEventRecord(c->ev_pool[slot].second, s); }
};

void prof_collect(rdmi_ctx* c) {
    for (size_t i = 0; i < c->ev_used; ++i) {
        hipEventSynchronize(c->ev_pool[i].second);
        float ms = 0;
        hipEventElapsedTime(&ms, c->ev_pool[i].first, c->ev_pool[i].second);
        ProfEntry& e = c->prof[(size_t)c->ev_entry[i]];
        e.ms += ms; e.launches += 1;
    }
    c->ev_used = 0;
}

template <int WM, int WN, int WK, int MT, int NT, int PF, bool BF16>
int launch_conv_t(const ConvArgs& a, hipStream_t s) {
    static bool attr_set = false;
    auto k = conv_mfma_kernel<WM, WN, WK, MT, NT, PF, BF16>;
    if (!attr_set) { HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); attr_set = true; }
    dim3 grid((unsigned)ceil_div(a.NB, a.S), (unsigned)(a.Cout_pad / a.BN));
    hipLaunchKernelGGL(k, grid, dim3(RDMI_THREADS), conv_lds_bytes(a), s, a);
    HIP_OK(hipGetLastError());
    return 0;
}

int launch_conv(int cfg, const ConvArgs& a_in, hipStream_t s) {
    ConvArgs a = a_in;
    // Small batches: a launch of one 64-column tile per sample leaves CUs idle (B = 128 at 9x9: 128 workgroups on 256 CUs).  The
    // <2,2,1,3,1> configuration computes the same rows with 32-column tiles -- twice the workgroups, same tables (S = 1, Mpad <= 96).
    static const int split_below = [] { const char* e = std::getenv("RDMI_CONV_SPLIT_BELOW"); return e ? atoi(e) : 200; }();
    if (cfg == 0 && a.S == 1 && a.Mpad <= 96 && (a.Cout_pad % 32) == 0 && ceil_div(a.NB, a.S) * (a.Cout_pad / a.BN) < split_below) { cfg = 1; a.BN = 32; }
    if (a.bf16) {
        if ((a.Cv & 31) || (a.Csc & 31)) return fail("bf16 conv launch with %d / %d input channels (multiples of 32 needed)", a.Cv, a.Csc);
        switch (cfg) {
            case 0: return launch_conv_t<1, 4, 1, 6, 1, 2, true>(a, s);
            case 1: return launch_conv_t<2, 2, 1, 3, 1, 3, true>(a, s);
            case 2: return launch_conv_t<1, 2, 2, 1, 1, 6, true>(a, s);
            case 3: return launch_conv_t<4, 1, 1, 2, 1, 4, true>(a, s);
        }
        return fail("bad conv cfg %d", cfg);
    }
    switch (cfg) {
        case 0: return launch_conv_t<1, 4, 1, 6, 1, 2, false>(a, s);
        case 1: return launch_conv_t<2, 2, 1, 3, 1, 3, false>(a, s);
        case 2: return launch_conv_t<1, 2, 2, 1, 1, 6, false>(a, s);
        case 3: return launch_conv_t<4, 1, 1, 2, 1, 4, false>(a, s);
    }
    return fail("bad conv cfg %d", cfg);
}

const char* cfg_name(int cfg) {
    static const char* n[] = {"conv_mfma<1,4,1,6,1>", "conv_mfma<2,2,1,3,1>", "conv_mfma<1,2,2,1,1>", "conv_mfma<4,1,1,2,1>"};
    return n[cfg];
}

int launch_attn(const AttnArgs& a, hipStream_t s) {
    static bool attr_set = false;
    auto k = attn_mfma_kernel<64>;
    if (!attr_set) { HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); attr_set = true; }
    hipLaunchKernelGGL(k, dim3((unsigned)a.NB), dim3(RDMI_THREADS), attn_lds_bytes<64>(a.Lpad, a.G), s, a);
    HIP_OK(hipGetLastError());
    return 0;
}

const float* P(rdmi_ctx* c, const std::string& name) { return c->params[(size_t)c->pindex.at(name)].ptr; }

int do_repack(rdmi_ctx* c, hipStream_t s) {
    for (size_t i = 0; i < c->jobs.size(); ++i) {
        const Param& p = c->params[(size_t)c->job_param[i]];
        if (!p.ptr) return fail("parameter '%s' was never bound (rdmi_set_param)", p.name.c_str());
        c->jobs[i].src = p.ptr;
        const int p2 = i < c->job_param2.size() ? c->job_param2[i] : -1;
        c->jobs[i].src2 = p2 >= 0 ? c->params[(size_t)p2].ptr : nullptr;
        if (p2 >= 0 && !c->jobs[i].src2) return fail("parameter '%s' was never bound (rdmi_set_param)", c->params[(size_t)p2].name.c_str());
    }
    for (auto& p : c->params)
        if (!p.ptr) return fail("parameter '%s' was never bound (rdmi_set_param)", p.name.c_str());
    HIP_OK(hipMemcpyAsync(c->d_jobs, c->jobs.data(), c->jobs.size() * sizeof(PackJob), hipMemcpyHostToDevice, s));
    {
        ProfScope ps(c, s, "pack_kernel", 0);
        hipLaunchKernelGGL(pack_kernel, dim3(32, (unsigned)c->jobs.size()), dim3(RDMI_THREADS), 0, s, (const PackJob*)c->d_jobs);
    }
    HIP_OK(hipGetLastError());
    // parameter-dependent pointers
    for (auto& op : c->ops) {
        if (op.kind == OP_CONV) {
            ConvArgs& a = op.conv;
            a.bias = P(c, op.p_bias);
            a.bias_sc = op.p_bias_sc.empty() ? nullptr : P(c, op.p_bias_sc);
            a.gamma = op.p_gamma.empty() ? nullptr : P(c, op.p_gamma);
            a.beta = op.p_beta.empty() ? nullptr : P(c, op.p_beta);
        } else {
            op.attn.gamma = P(c, op.p_gamma); op.attn.beta = P(c, op.p_beta); op.attn.b3 = P(c, op.p_b3);
        }
    }
    for (auto& l : c->tl) {
        if (l.kind == 6) { l.gact.gamma = P(c, l.p_gamma); l.gact.beta = P(c, l.p_beta); continue; }
        if (l.kind != 0) continue;
        l.conv.bias = l.p_bias.empty() ? c->d_w + l.bias_arena : P(c, l.p_bias);
        l.conv.gamma = l.p_gamma.empty() ? nullptr : P(c, l.p_gamma);
        l.conv.beta = l.p_beta.empty() ? nullptr : P(c, l.p_beta);
    }
    for (auto& q : c->progs) {
        if (!q.ok) continue;
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
    }
    if (c->fused_ready()) HIP_OK(hipStreamSynchronize(s));
    c->packed_valid = true;
    return 0;
}

struct FwdIn {
    const float* x; int x_mod;            // x holds x_mod samples; sample n reads x[n % x_mod] (0: n)
    const float* sig; int sig_mod;        // sigma or t per sample (n % sig_mod); null: every sample uses t_scalar
    float t_scalar;
    int t_is_time; float smin, ratio;
    const float* labels; int label_rows;
    float* out; int NB;
    const float* tt_row = nullptr;        // sampler: precomputed time_mlp.2 output row (+ both biases) of this evaluation's t
    const float* dense_rows = nullptr;    // sampler: precomputed Dense_0 outputs [NB][dense_total] of this evaluation (skips the whole embedding)
};

int run_forward(rdmi_ctx* c, const FwdIn& f, hipStream_t s) {
    const rdmi_arch& a = c->arch;
    if (f.NB < 1 || f.NB > c->max_batch) return fail("batch %d outside [1, %d] (rdmi_create max_batch)", f.NB, c->max_batch);
    if (a.scale_by_sigma && !c->tiled) return fail("scale_by_sigma=True is only built for the tiled plan (the shipped GTO-Halo config sets it False, RD/configs/model/ncsnpp.yaml)");
    if (a.conditional && !f.labels) return fail("class_labels is required: the model is conditional (label_emb) -- reference raises here too (RD/models/ncsnpp.py:262)");
    const int T = c->temb, Np_t = (T + 63) & ~63, Np_d = (c->dense_total + 63) & ~63;
    // ---- embedding: Fourier -> Linear -> SiLU -> Linear (+label_emb) -> [SiLU -> all Dense_0]
    if (!f.tt_row && !f.dense_rows) {
    LinArgs l{};
    l.M = f.NB; l.fourW = P(c, "time_embed.W"); l.nfour = a.nf;
    l.X = f.sig; l.x_mod = f.sig_mod > 0 ? f.sig_mod : f.NB; l.t_is_time = f.t_is_time; l.smin = f.smin; l.ratio = f.ratio;
    l.use_scalar = f.sig ? 0 : 1; l.t_scalar = f.t_scalar;
    l.pre = 2; l.K = pad16(2 * a.nf); l.W = c->d_w + c->w_t0; l.Npad = Np_t; l.N = T; l.bias = P(c, "time_mlp.0.bias");
    l.Y = c->d_h1; l.ldy = T;
    {
        ProfScope ps(c, s, "linear_mfma(time_mlp.0)", 2.0 * f.NB * T * 2 * a.nf);
        hipLaunchKernelGGL(linear_mfma_kernel<1>, dim3((unsigned)ceil_div(f.NB, 16), (unsigned)(Np_t / 64)), dim3(RDMI_THREADS), 0, s, l);
    }
    LinArgs l2{};
    l2.M = f.NB; l2.X = c->d_h1; l2.ldx = T; l2.pre = 1; l2.K = T; l2.W = c->d_w + c->w_t2; l2.Npad = Np_t; l2.N = T;
    l2.bias = P(c, "time_mlp.2.bias"); l2.Y = c->d_temb; l2.ldy = T;
    if (a.conditional) { l2.labels = f.labels; l2.Wl = P(c, "label_emb.weight"); l2.bl = P(c, "label_emb.bias"); l2.ncls = a.num_classes; l2.label_rows = f.label_rows; }
    {
        ProfScope ps(c, s, "linear_mfma(time_mlp.2)", 2.0 * f.NB * T * T);
        hipLaunchKernelGGL(linear_mfma_kernel<1>, dim3((unsigned)ceil_div(f.NB, 16), (unsigned)(Np_t / 64)), dim3(RDMI_THREADS), 0, s, l2);
    }
    }
    if (!f.dense_rows) {
    LinArgs l3{};
    l3.M = f.NB; l3.X = c->d_temb; l3.ldx = T; l3.pre = 1; l3.K = T;
    if (f.tt_row) {     // every sample shares the time row; the label embedding is added in the prologue (same sums as time_mlp.2's epilogue)
        l3.X = f.tt_row; l3.ldx = 0; l3.pre = 3;
        if (a.conditional) { l3.labels = f.labels; l3.Wl = P(c, "label_emb.weight"); l3.ncls = a.num_classes; l3.label_rows = f.label_rows; }
    } l3.W = c->d_w + c->w_dense; l3.Npad = Np_d; l3.N = c->dense_total;
    l3.bias = c->d_w + c->b_dense; l3.Y = c->d_dense; l3.ldy = c->dense_total;
    {
        ProfScope ps(c, s, "linear_mfma(Dense_0 x all)", 2.0 * f.NB * T * c->dense_total);
        hipLaunchKernelGGL((linear_mfma_kernel<1, 16>), dim3((unsigned)ceil_div(f.NB, 16), (unsigned)(Np_d / 64)), dim3(RDMI_THREADS), 0, s, l3);
    }
    }
    HIP_OK(hipGetLastError());
    if (a.compute_dtype == 1 && !c->tiled && !c->in_train_forward)
        return fail("compute_dtype=bf16 on this shape is built for the TRAINING step only (model.train_dtype = 'bf16'); sampling and evaluation of the "
                    "GTO-Halo model run the exact-fp32 plans");
    if (c->tiled) return run_tiled(c, f.x, f.x_mod, f.sig, f.sig_mod, f.t_is_time, f.t_scalar, f.smin, f.ratio, f.out, f.NB, f.dense_rows, s);
    // ---- the U-Net: one workgroup-resident launch (csrc/unet_kernel.h) ...
    const rdmi_ctx::FusedProg* fq = (c->use_fused && !c->debug_taps) ? c->pick(f.NB) : nullptr;
    if (c->in_train_forward) {      // one sample per workgroup pays while the batch fits the chip in about one wave of workgroups; beyond that the layer plan (bf16 MFMA) is faster
        static const int fused_upto = [] { const char* e = std::getenv("RDMI_TRAIN_FUSED_UPTO"); return e ? atoi(e) : 2 * std::max(resident_workgroups(), 1); }();
        fq = (c->train_prog >= 0 && c->progs[(size_t)c->train_prog].ok && f.NB <= fused_upto) ? &c->progs[(size_t)c->train_prog] : nullptr;
    }
    if (fq) {
        static bool attr_set = false;
        if (!attr_set) {
            HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(unet_wg_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(unet_wg_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>((unet_wg_kernel<false, true>)), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>((unet_wg_kernel<false, true, true>)), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>((unet_wg_kernel<true, true, true>)), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>((unet_wg_kernel<false, false, false, true>)), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            attr_set = true;
        }
        UnetArgs ua = fq->fargs;
        ua.x_in = f.x; ua.x_mod = f.x_mod; ua.out = f.out; ua.NB = f.NB;
        unsigned nwg = (unsigned)ceil_div(f.NB, fq->S);
        if (fq->coop) {          // groups of four workgroups, whole blocks of 4 * stride ids; every launch gets fresh exchange tags
            nwg = (unsigned)(ceil_div(ceil_div(f.NB, 4), ua.coop_stride) * 4 * ua.coop_stride);
            ua.epoch_base = c->coop_epoch;
            c->coop_epoch += (unsigned)fq->n_xchg;
            if (c->coop_epoch > 0xfff00000u) c->coop_epoch = 0;        // tags only have to differ from the two previous uses of a slot
        }
        if (f.dense_rows) ua.dense = f.dense_rows;
        double fl = 0;
        for (auto& op : c->ops) fl += op.flops_per_sample;
        ProfScope ps(c, s, "unet_wg_kernel", fl * f.NB);
        c->last_prog = (int)(fq - c->progs.data());
        if (fq->train) {
            ua.drop_p = c->train_drop_p; ua.seed_dev = c->train_seed_dev;
            hipLaunchKernelGGL((unet_wg_kernel<false, false, false, true>), dim3(nwg), dim3(UW_THREADS), fq->fused_lds, s, ua);
            HIP_OK(hipGetLastError());
            return 0;
        }
        if ((ua.stamps || ua.dbg) && fq->S == 1) hipLaunchKernelGGL(unet_wg_kernel<true>, dim3(nwg), dim3(UW_THREADS), fq->fused_lds, s, ua);   // diagnostic build
        else if ((ua.stamps || ua.dbg) && fq->coop) hipLaunchKernelGGL((unet_wg_kernel<true, true, true>), dim3(nwg), dim3(UW_THREADS), fq->fused_lds, s, ua);
        else if (fq->coop) hipLaunchKernelGGL((unet_wg_kernel<false, true, true>), dim3(nwg), dim3(UW_THREADS), fq->fused_lds, s, ua);
        else if (fq->S > 1) hipLaunchKernelGGL((unet_wg_kernel<false, true>), dim3(nwg), dim3(UW_THREADS), fq->fused_lds, s, ua);
        else hipLaunchKernelGGL(unet_wg_kernel<false>, dim3(nwg), dim3(UW_THREADS), fq->fused_lds, s, ua);
        HIP_OK(hipGetLastError());
        return 0;
    }
    // ---- ... or the layer-by-layer plan (debug taps, shapes the fused planner cannot fit in LDS)
    for (auto& op : c->ops) {
        if (op.kind == OP_CONV) {
            ConvArgs ca = op.conv;
            ca.NB = f.NB;
            if (op.a_is_input) { ca.srcA = f.x; ca.srcA_mod = f.x_mod; }
            if (op.out_is_output) ca.out = f.out;
            if (op.use_dense && f.dense_rows) ca.dense = f.dense_rows;
            ProfScope ps(c, s, cfg_name(op.cfg), op.flops_per_sample * f.NB);
            if (int e = launch_conv(op.cfg, ca, s)) return e;
        } else {
            AttnArgs aa = op.attn;
            aa.NB = f.NB;
            ProfScope ps(c, s, "attn_mfma<64>", op.flops_per_sample * f.NB);
            if (int e = launch_attn(aa, s)) return e;
        }
    }
    return 0;
}


int run_tiled(rdmi_ctx* c, const float* x, int x_mod, const float* sig, int sig_mod, int sig_is_time, float t_scalar, float smin, float ratio, float* out, int NB,
              const float* dense_rows, hipStream_t s) {
    const rdmi_arch& a = c->arch;
    const int HW = c->H * c->W, Cc = a.channels;
    const long tot = (long)NB * HW * Cc;
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3((unsigned)((tot + RDMI_THREADS - 1) / RDMI_THREADS)), dim3(RDMI_THREADS), 0, s, x, c->t_xin, NB, HW, Cc, x_mod);
    for (auto& l : c->tl) {
        if (l.kind == 0) {
            TConvArgs ca = l.conv;
            ca.NB = NB;
            if (ca.dense && dense_rows) ca.dense = dense_rows;        // the sampler's per-update Dense_0 rows
            if (l.out_is_final && a.scale_by_sigma) {               // h / time_cond (RD/models/ncsnpp.py:350-351)
                if (sig) { ca.sig = sig; ca.sig_mod = sig_mod; ca.sig_is_time = sig_is_time; ca.smin = smin; ca.ratio = ratio; }
                else ca.out_scale /= sig_is_time ? smin * powf(ratio, t_scalar) : t_scalar;
            }
            const unsigned tiles = (unsigned)ceil_div(ca.Ho, ca.TR);
            // column tiles per wave (the workgroup covers 64 * nct channels: staging and its GroupNorm/SiLU arithmetic are shared), as
            // long as the launch still has >= 2 workgroups per CU
            int nct = 1;
            for (int cand : {4, 2})
                if (ca.Cout_pad >= 64 * cand && (long)tiles * NB * ceil_div(ca.Cout_pad, 64 * cand) >= c->tiled_min_wgs) { nct = cand; break; }
            dim3 grid(tiles * (unsigned)NB, (unsigned)ceil_div(ca.Cout_pad, 64 * nct));
            const bool h = a.compute_dtype == 1;
            const size_t lds = l.pre ? tconv_pre_lds_bytes(ca) : h ? tconv_bf16_lds_bytes(ca) : tconv_lds_bytes(ca);
#define RDMI_TCONV_PRE(NMT_, NCT_)                                                                                                    \
    do {                                                                                                                              \
        if (ca.ntap == 1) hipLaunchKernelGGL((tconv_pre_kernel<NMT_, NCT_, 1>), grid, dim3(RDMI_THREADS), lds, s, ca);                \
        else hipLaunchKernelGGL((tconv_pre_kernel<NMT_, NCT_, 9>), grid, dim3(RDMI_THREADS), lds, s, ca);                             \
    } while (0)
            // implicit-GEMM form (iconv_kernel: 128 x 128 tiles over the whole batch's pixels) once the launch has a workgroup per CU
            // (read at context creation -- off by default, see finish_tiled_plan; RDMI_ICONV_MIN_WGS: tests set the threshold to reach the kernel at fixture batches)
            if (l.pre && ca.stride == 1 && !ca.up && ca.TR * ca.Wo == 64 && (ca.Ho * ca.Wo) % 64 == 0 && ca.Cout % 128 == 0 && ca.Cv % 64 == 0) {
                const long M = (long)NB * ca.Ho * ca.Wo;
                const dim3 g2((unsigned)((M + 127) / 128), (unsigned)(ca.Cout / 128));
                if ((long)g2.x * g2.y >= c->iconv_min_wgs) {
                    static const bool by_shape = std::getenv("RDMI_PROF_SHAPES") != nullptr;      // diagnostic: one profile row per conv shape
                    ProfScope ps2(c, s, by_shape ? "iconv_kernel<bf16> " + std::to_string(ca.Wo) + "x" + std::to_string(ca.Ho) + " K" + std::to_string(ca.ntap * ca.Cv) + " N" + std::to_string(ca.Cout) + (ca.resid ? " +res" : "") : std::string("iconv_kernel<bf16>"), l.flops_per_sample * NB);
                    if (ca.ntap == 1) hipLaunchKernelGGL((iconv_kernel<1>), g2, dim3(RDMI_THREADS), iconv_lds_bytes(), s, ca);
                    else hipLaunchKernelGGL((iconv_kernel<9>), g2, dim3(RDMI_THREADS), iconv_lds_bytes(), s, ca);
                    continue;
                }
            }
            ProfScope ps(c, s, l.pre ? "tconv_pre_kernel<bf16>" : h ? "tconv_kernel<bf16>" : "tconv_kernel<fp32>", l.flops_per_sample * NB);
            if (l.pre) {
                if (l.nmt == 4) { if (nct == 4) RDMI_TCONV_PRE(4, 4); else if (nct == 2) RDMI_TCONV_PRE(4, 2); else RDMI_TCONV_PRE(4, 1); }
                else { if (nct == 4) RDMI_TCONV_PRE(1, 4); else if (nct == 2) RDMI_TCONV_PRE(1, 2); else RDMI_TCONV_PRE(1, 1); }
                continue;
            }
#undef RDMI_TCONV_PRE
#define RDMI_TCONV(NMT_, NCT_)                                                                                                        \
    do {                                                                                                                              \
        if (h) hipLaunchKernelGGL((tconv_kernel<NMT_, NCT_, true>), grid, dim3(RDMI_THREADS), lds, s, ca);                            \
        else hipLaunchKernelGGL((tconv_kernel<NMT_, NCT_, false>), grid, dim3(RDMI_THREADS), lds, s, ca);                             \
    } while (0)
            if (l.nmt == 4) { if (nct == 4) RDMI_TCONV(4, 4); else if (nct == 2) RDMI_TCONV(4, 2); else RDMI_TCONV(4, 1); }
            else { if (nct == 4) RDMI_TCONV(1, 4); else if (nct == 2) RDMI_TCONV(1, 2); else RDMI_TCONV(1, 1); }
#undef RDMI_TCONV
        } else if (l.kind == 1) {
            ProfScope ps(c, s, "gn_stats_kernel", 0);
            hipLaunchKernelGGL(gn_stats_kernel, dim3((unsigned)l.G, (unsigned)NB), dim3(RDMI_THREADS), 16, s, l.sA, l.sB, l.CA, l.CB, l.HW, l.G, 1e-6f, l.stats);
        } else if (l.kind == 2) {
            ProfScope ps(c, s, a.compute_dtype == 1 ? "bgemm_nt_bf16_kernel" : "bgemm_nt_kernel", l.flops_per_sample * NB);
            if (a.compute_dtype == 1) hipLaunchKernelGGL(bgemm_nt_bf16_kernel, dim3((unsigned)ceil_div(l.gemm.M, 64), (unsigned)ceil_div(l.gemm.N, 64), (unsigned)NB), dim3(RDMI_THREADS), 0, s, l.gemm);
            else hipLaunchKernelGGL(bgemm_nt_kernel, dim3((unsigned)ceil_div(l.gemm.M, 64), (unsigned)ceil_div(l.gemm.N, 64), (unsigned)NB), dim3(RDMI_THREADS), 0, s, l.gemm);
        } else if (l.kind == 7) {
            FlashArgs fa = l.flash; fa.NB = NB;
            ProfScope ps(c, s, "flash_attn_bf16_kernel", l.flops_per_sample * NB);
            const dim3 grid((unsigned)(fa.L / 64), (unsigned)NB);
            static bool flash_attr = false;                          // 70 KB of dynamic LDS at C = 256
            if (!flash_attr) {
                HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(flash_attn_bf16_kernel<256>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(flash_attn_bf16_kernel<128>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                flash_attr = true;
            }
            if (l.flashC == 256) hipLaunchKernelGGL(flash_attn_bf16_kernel<256>, grid, dim3(RDMI_THREADS), flash_lds_bytes<256>(), s, fa);
            else if (l.flashC == 128) hipLaunchKernelGGL(flash_attn_bf16_kernel<128>, grid, dim3(RDMI_THREADS), flash_lds_bytes<128>(), s, fa);
            else hipLaunchKernelGGL(flash_attn_bf16_kernel<64>, grid, dim3(RDMI_THREADS), flash_lds_bytes<64>(), s, fa);
        } else if (l.kind == 6) {
            GnActArgs g = l.gact; g.NB = NB;
            if (l.fin) {
                ProfScope ps(c, s, "gn_act_fin_kernel", 0);
                hipLaunchKernelGGL(gn_act_fin_kernel, dim3((unsigned)ceil_div(g.HW, GA_PIX), (unsigned)(g.Cv / 64), (unsigned)NB), dim3(RDMI_THREADS), gn_act_fin_lds_bytes(std::max(g.tilesA, g.tilesB)), s, g);
                continue;
            }
            const long units = (long)NB * g.HW * (g.Cv / 8);
            ProfScope ps(c, s, "gn_act_kernel", 0);
            hipLaunchKernelGGL(gn_act_kernel, dim3((unsigned)((units + RDMI_THREADS - 1) / RDMI_THREADS)), dim3(RDMI_THREADS), 0, s, g);
        } else if (l.kind == 5) {
            ProfScope ps(c, s, "gn_finalize_kernel", 0);
            hipLaunchKernelGGL(gn_finalize_kernel, dim3((unsigned)NB), dim3(RDMI_THREADS), 0, s, l.sA, l.sB, l.CA, l.CB, l.tL, l.tC, l.pxA, l.pxB, l.HW, l.G, 1e-6f, l.stats);
        } else if (l.kind == 3) {
            const long rows = l.rows_per_sample * NB;
            ProfScope ps(c, s, "softmax_rows_kernel", 0);
            hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(RDMI_THREADS), 0, s, l.sm, rows, l.L);
        } else {
            const long n = (long)NB * l.tL * l.tC;
            ProfScope ps(c, s, "transpose_lc_kernel", 0);
            if (l.t16 && l.tL % 64 == 0 && l.tC % 64 == 0)
                hipLaunchKernelGGL(transpose_lc_tile16_kernel, dim3((unsigned)(l.tL / 64), (unsigned)(l.tC / 64), (unsigned)NB), dim3(RDMI_THREADS), 0, s,
                                   reinterpret_cast<const bf16_t*>(l.tsrc), reinterpret_cast<bf16_t*>(l.tdst), l.tL, l.tC, l.tld, l.tc0);
            else if (l.t16) return fail("bf16 q | k | v: the transpose needs L and C in multiples of 64");
            else if (l.tL % 64 == 0 && l.tC % 64 == 0)
                hipLaunchKernelGGL(transpose_lc_tile_kernel, dim3((unsigned)(l.tL / 64), (unsigned)(l.tC / 64), (unsigned)NB), dim3(RDMI_THREADS), 0, s, l.tsrc, l.tdst, l.tL, l.tC, l.tld, l.tc0);
            else
            hipLaunchKernelGGL(transpose_lc_kernel, dim3((unsigned)((n + RDMI_THREADS - 1) / RDMI_THREADS)), dim3(RDMI_THREADS), 0, s, l.tsrc, l.tdst, NB, l.tL, l.tC, l.tld, l.tc0);
        }
    }
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3((unsigned)((tot + RDMI_THREADS - 1) / RDMI_THREADS)), dim3(RDMI_THREADS), 0, s, (const float*)c->t_out, out, NB, HW, Cc, 0);
    HIP_OK(hipGetLastError());
    return 0;
}

}  // namespace

#include "train_plan.h"

// ============================================================================================
// C ABI
// ============================================================================================
extern "C" {

const char* rdmi_last_error(void) { return g_err.c_str(); }
int rdmi_debug_op_cycles(rdmi_ctx* c, long long* host, int cap, const char** desc, int desc_cap) {
    if (!c || !c->d_stamps || !c->fused_ready()) return 0;
    const rdmi_ctx::FusedProg& q0 = c->progs[(size_t)std::min<int>(std::max(c->last_prog, 0), (int)c->progs.size() - 1)];     // the program of the last launch
    const int n = (int)q0.fprog.size();
    std::vector<long long> st((size_t)n + 1);
    if (n > 1000) return 0;
    if (hipMemcpy(st.data(), c->d_stamps, st.size() * sizeof(long long), hipMemcpyDeviceToHost) != hipSuccess) return 0;
    for (int i = 0; i < n && i < cap; ++i) host[i] = st[(size_t)i + 1] - st[(size_t)i];
    // fine stamps of CONV ops (entry, ring issued, first B landed, main done, shortcut done, epilogue done) follow at cap/2
    {
        std::vector<long long> fs((size_t)n * 8);
        if (hipMemcpy(fs.data(), c->d_stamps + 1024, fs.size() * sizeof(long long), hipMemcpyDeviceToHost) == hipSuccess)
            for (int i = 0; i < n && 256 + i * 8 + 7 < cap; ++i)
                for (int k = 0; k < 8; ++k) host[256 + i * 8 + k] = fs[(size_t)i * 8 + k] - st[(size_t)i];
    }
    for (int i = 0; i < n && i < desc_cap; ++i) desc[i] = q0.fdesc[(size_t)i].c_str();
    return n;
}

const char* rdmi_path_info(rdmi_ctx* c) {
    static thread_local std::string s;
    if (!c) return "";
    if (c->tiled) {
        s = std::string("tiled (") + (c->arch.compute_dtype == 1 ? "bf16 MFMA operands, fp32 accumulate" : "fp32") + "): " + std::to_string(c->tl.size()) + " launches over HBM-resident NHWC tensors (" + std::to_string(c->t_ws_per_sample * 4 / 1024) + " KiB of activations per sample)";
    } else if (c->fused_ready() && c->use_fused && !c->debug_taps) {
        s = "fused: workgroup-resident U-Net, " + std::to_string(c->progs[0].fprog.size()) + " ops, " + std::to_string(c->progs[0].fused_lds) + " B LDS";
        for (size_t i = 1; i < c->progs.size(); ++i)
            if (c->progs[i].coop)
                s += c->progs[i].ok ? "; co-operative groups of 4 workgroups up to batch " + std::to_string(c->progs[i].max_nb) + " (" + std::to_string(c->progs[i].fprog.size()) + " ops, " + std::to_string(c->progs[i].n_xchg) + " exchanges, id stride " + std::to_string(c->coop_stride) + (c->use_coop ? ")" : ", disabled)")
                                    : "; co-operative program unavailable (" + c->progs[i].why + ")";
            else
            s += c->progs[i].ok ? "; S=" + std::to_string(c->progs[i].S) + " samples/workgroup from batch " + std::to_string(c->s_min_wg * c->progs[i].S) + " (" + std::to_string(c->progs[i].fprog.size()) + " ops)"
                                : "; S=" + std::to_string(c->progs[i].S) + " unavailable (" + c->progs[i].why + ")";
    } else {
        s = "layers: " + std::to_string(c->ops.size()) + " launches" + (c->fused_ready() ? "" : " (fused unavailable: " + (c->progs.empty() ? std::string("not built") : c->progs[0].why) + ")");
    }
    for (auto& q : c->progs) if (q.train && !q.ok) s += "; fused training forward unavailable (" + q.why + ")";
    if (c->train_prog >= 0 && c->progs[(size_t)c->train_prog].ok)
        s += "; training forward: fused program (" + std::to_string(c->progs[(size_t)c->train_prog].fprog.size()) + " ops, layer outputs stashed for the backward)";
    return s.c_str();
}

int rdmi_philox_normal(float* z, size_t n, unsigned long long seed, unsigned long long elem_offset, unsigned draw, void* stream) {
    if (n == 0) return 0;
    if (!z) return fail("null argument");
    hipLaunchKernelGGL(philox_normal_kernel, dim3((unsigned)((n + 4 * RDMI_THREADS - 1) / (4 * RDMI_THREADS))), dim3(RDMI_THREADS), 0, (hipStream_t)stream, z, (long)n,
                       (uint64_t)seed, (uint64_t)elem_offset, (const int*)nullptr, (int)draw);
    HIP_OK(hipGetLastError());
    return 0;
}

// 0: no co-operative exchange of this context has ever given up its bounded wait; 1: one did (the affected output samples are NaN).
// Synchronises the device (diagnostic: tests and bench.py call it after their timed regions).
int rdmi_coop_status(rdmi_ctx* c, int* gave_up) {
    if (!c || !gave_up) return fail("null argument");
    *gave_up = 0;
    for (auto& q : c->progs)
        if (q.coop && q.d_coop_err) { int v = 0; HIP_OK(hipMemcpy(&v, q.d_coop_err, sizeof v, hipMemcpyDeviceToHost)); *gave_up |= v; }
    if (*gave_up) c->use_coop = false;       // the groups of this device were not co-resident once: later calls take the single-sample programs
    return 0;
}
const char* rdmi_version(void) {
#ifdef RDMI_EMU
    return "rdmi 0.1 (CPU execution-model emulator build: tests only)";
#else
    return "rdmi 0.1 (gfx950)";
#endif
}

int rdmi_create(const rdmi_arch* arch, int max_batch, int H, int W, rdmi_ctx** out) {
    if (!arch || !out) return fail("null argument");
    if (arch->n_levels < 1 || arch->n_levels > RDMI_MAX_LEVELS) return fail("n_levels=%d out of range", arch->n_levels);
    if (arch->nf % 16 != 0) return fail("nf=%d must be a multiple of 16", arch->nf);
    if (max_batch < 1 || H < 2 || W < 2) return fail("bad shape max_batch=%d H=%d W=%d", max_batch, H, W);
    rdmi_ctx* c = new rdmi_ctx();
    c->arch = *arch;
    c->max_batch = max_batch; c->H = H; c->W = W;
    c->temb = arch->nf * 4;
    if (const char* e = std::getenv("RDMI_DEBUG_TAPS")) c->debug_taps = atoi(e) != 0;
    if (const char* e = std::getenv("RDMI_PATH")) c->use_fused = std::string(e) != "layers";
    int e = 0;
    try { e = build_plan(c); } catch (const std::exception& ex) { e = fail("plan construction failed: %s", ex.what()); }
    if (e) { rdmi_destroy(c); return e; }
    *out = c;
    return 0;
}

int rdmi_destroy(rdmi_ctx* c) {
    if (!c) return 0;
    if (TrainPlan* T = get_train(c)) {
        if (T->side) { (void)hipStreamSynchronize(T->side); (void)hipStreamDestroy(T->side); }
        for (int p = 0; p < TrainPlan::MAXSETS; ++p) for (void* q : {(void*)T->G[p], (void*)T->ACT[p], (void*)T->ACTS[p]}) if (q) (void)hipFree(q);
        for (int p = 0; p < 2; ++p) {
            if (T->ev_ready[p]) (void)hipEventDestroy(T->ev_ready[p]);
            if (T->ev_done[p]) (void)hipEventDestroy(T->ev_done[p]);
        }
        if (T->fwd_exec) (void)hipGraphExecDestroy(T->fwd_exec);
        if (T->bwd_exec) (void)hipGraphExecDestroy(T->bwd_exec);
        if (T->cap) (void)hipStreamDestroy(T->cap);
        if (T->h_seed) (void)hipHostFree(T->h_seed);
        for (void* p : {(void*)T->x_in, (void*)T->out_buf, (void*)T->gout_buf, (void*)T->grads_int, (void*)T->d_seed}) if (p) (void)hipFree(p);
        for (void* p : {(void*)T->d_jobs, (void*)T->d_wb, (void*)T->d_int, (void*)T->gws, (void*)T->GA, (void*)T->GS, (void*)T->zero_bias, (void*)T->gdense, (void*)T->gta, (void*)T->gh1, (void*)T->four, (void*)T->sig_copy, (void*)T->lab_copy,
                        (void*)T->d_gemm_jobs, (void*)T->d_col_jobs}) if (p) (void)hipFree(p);
        train_registry().erase(c);
        delete T;
    }
    for (auto& q : c->progs) for (void* p : {(void*)q.d_fprog, (void*)q.d_ftabs, (void*)q.d_spill, (void*)q.d_xbuf, (void*)q.d_coop_err}) if (p) hipFree(p);
    void* ptrs[] = {c->d_fprog, c->d_ftabs, c->d_spill, c->d_jobs, c->d_w, c->d_int, c->ws, c->d_h1, c->d_temb, c->d_dense, c->d_s2, c->d_score, c->d_z, c->d_norms, c->d_ts, c->d_tvec, c->d_state, c->d_tt, c->d_th1, c->d_dense_all, c->t_ws, c->t_xin, c->t_out, c->d_w16, c->t_zero};
    for (void* p : ptrs) if (p) hipFree(p);
    for (auto& ev : c->ev_pool) { hipEventDestroy(ev.first); hipEventDestroy(ev.second); }
    delete c;
    return 0;
}

int rdmi_num_params(const rdmi_ctx* c) { return c ? (int)c->params.size() : 0; }

int rdmi_param_info(const rdmi_ctx* c, int index, const char** name, size_t* numel) {
    if (!c || index < 0 || index >= (int)c->params.size()) return fail("param index %d out of range", index);
    if (name) *name = c->params[(size_t)index].name.c_str();
    if (numel) *numel = c->params[(size_t)index].numel;
    return 0;
}

int rdmi_set_param(rdmi_ctx* c, const char* name, const float* dev_ptr, size_t numel) {
    if (!c || !name || !dev_ptr) return fail("null argument");
    auto it = c->pindex.find(name);
    if (it == c->pindex.end()) return fail("unknown parameter '%s'", name);
    Param& p = c->params[(size_t)it->second];
    if (p.numel != numel) return fail("parameter '%s': expected %zu elements, got %zu", name, p.numel, numel);
    if (p.ptr != dev_ptr) c->packed_valid = false;
    p.ptr = dev_ptr;
    return 0;
}

int rdmi_repack(rdmi_ctx* c, void* stream) {
    if (!c) return fail("null context");
    return do_repack(c, (hipStream_t)stream);
}

static int maybe_repack(rdmi_ctx* c, unsigned flags, hipStream_t s) {
    if ((flags & RDMI_PARAMS_CACHED) && c->packed_valid) return 0;
    return do_repack(c, s);
}

int rdmi_forward(rdmi_ctx* c, const float* x, const float* sigma, const float* labels, float* out, int B,
                 unsigned flags, void* stream) {
    if (!c || !x || !sigma || !out) return fail("null argument");
    hipStream_t s = (hipStream_t)stream;
    if (int e = maybe_repack(c, flags, s)) return e;
    FwdIn f{x, 0, sigma, 0, 0.f, 0, 0.f, 0.f, labels, B, out, B};
    int e = run_forward(c, f, s);
    if (c->profiling) prof_collect(c);
    return e;
}

int rdmi_score(rdmi_ctx* c, const float* x, const float* t, const float* labels, float* out, int B,
               double sigma_min, double sigma_max, unsigned flags, void* stream) {
    if (!c || !x || !t || !out) return fail("null argument");
    hipStream_t s = (hipStream_t)stream;
    if (int e = maybe_repack(c, flags, s)) return e;
    FwdIn f{x, 0, t, 0, 0.f, 1, (float)sigma_min, (float)(sigma_max / sigma_min), labels, B, out, B};
    int e = run_forward(c, f, s);
    if (c->profiling) prof_collect(c);
    return e;
}

static int cf_score_impl(rdmi_ctx* c, const float* x, const float* t, int t_mod, const float* labels, const float* weight,
                         float* out, int B, float smin, float ratio, hipStream_t s) {
    const int E = c->H * c->W * c->arch.channels;
    FwdIn f{x, B, t, t_mod, 0.f, 1, smin, ratio, labels, B, c->d_s2, 2 * B};
    if (int e = run_forward(c, f, s)) return e;
    {
        ProfScope ps(c, s, "cfg_combine", 0);
        hipLaunchKernelGGL(cfg_combine_kernel, dim3((unsigned)ceil_div(B * E, RDMI_THREADS)), dim3(RDMI_THREADS), 0, s,
                           (const float*)c->d_s2, weight, out, B, E);
    }
    HIP_OK(hipGetLastError());
    return 0;
}

int rdmi_cf_score(rdmi_ctx* c, const float* x, const float* t, const float* labels, const float* weight, float* out,
                  int B, double sigma_min, double sigma_max, unsigned flags, void* stream) {
    if (!c || !x || !t || !labels || !out) return fail("null argument");
    if (2 * B > c->max_batch) return fail("CFG needs a model batch of 2*B=%d > max_batch=%d", 2 * B, c->max_batch);
    hipStream_t s = (hipStream_t)stream;
    if (int e = maybe_repack(c, flags, s)) return e;
    int e = cf_score_impl(c, x, t, B, labels, weight, out, B, (float)sigma_min, (float)(sigma_max / sigma_min), s);
    if (c->profiling) prof_collect(c);
    return e;
}

int rdmi_reflect(const float* in, float* out, size_t n, void* stream) {
    if (!in || !out) return fail("null argument");
    hipLaunchKernelGGL(reflect_kernel, dim3((unsigned)((n + RDMI_THREADS - 1) / RDMI_THREADS)), dim3(RDMI_THREADS), 0,
                       (hipStream_t)stream, in, out, (long)n);
    HIP_OK(hipGetLastError());
    return 0;
}

int rdmi_score_hk(const float* x, const float* x_orig, const float* sigma, float* out, int B, int E, int efs, int refls,
                  float min_cutoff, void* stream) {
    if (!x || !x_orig || !sigma || !out) return fail("null argument");
    hipLaunchKernelGGL(score_hk_kernel, dim3((unsigned)ceil_div(B * E, RDMI_THREADS)), dim3(RDMI_THREADS), 0, (hipStream_t)stream,
                       x, x_orig, sigma, out, B, E, efs, refls, min_cutoff);
    HIP_OK(hipGetLastError());
    return 0;
}


static float g_const(double smin, double smax) {
    // torch.sqrt(torch.tensor(2 * (np.log(sigma_max) - np.log(sigma_min)), dtype=float32))   RD/sde_lib.py:138-139
    return sqrtf((float)(2.0 * (std::log(smax) - std::log(smin))));
}

int rdmi_perturb(const float* batch, const float* z, const float* t, float* out, int B, int E, double sigma_min,
                 double sigma_max, void* stream) {
    if (!batch || !z || !t || !out) return fail("null argument");
    hipLaunchKernelGGL(perturb_kernel, dim3((unsigned)ceil_div(B * E, RDMI_THREADS)), dim3(RDMI_THREADS), 0, (hipStream_t)stream, batch,
                       z, t, out, B, E, (float)sigma_min, (float)(sigma_max / sigma_min));
    HIP_OK(hipGetLastError());
    return 0;
}

int rdmi_gto_pack(const float* data, const long long* idx, float* images, float* labels, int B, int row_len, int elems, double mean,
                  double std, void* stream) {
    if (B == 0) return 0;
    if (!data || !images || !labels) return fail("null argument");
    if (B < 0 || row_len < 1 || elems < row_len) return fail("gto_pack: rows of %d values into images of %d", row_len, elems);
    hipLaunchKernelGGL(gto_pack_kernel, dim3((unsigned)ceil_div(B * elems, RDMI_THREADS)), dim3(RDMI_THREADS), 0, (hipStream_t)stream, data, idx, images,
                       labels, B, row_len, elems, (float)mean, (float)std);
    HIP_OK(hipGetLastError());
    return 0;
}

int rdmi_gto_unnormalize(const float* samples, float* out, unsigned long long* clip_count, int N, int row_elems, void* stream) {
    if (N == 0) return 0;
    if (!samples || !out) return fail("null argument");
    if (N < 0 || row_elems < 67) return fail("gto_unnormalize: rows of %d values (need >= 67)", row_elems);
    hipLaunchKernelGGL(gto_unnormalize_kernel, dim3((unsigned)ceil_div(N * 22, RDMI_THREADS)), dim3(RDMI_THREADS), 0, (hipStream_t)stream, samples, out,
                       clip_count, N, row_elems);
    HIP_OK(hipGetLastError());
    return 0;
}

int rdmi_sm_loss(const float* score, const float* perturbed, const float* batch, const float* t, float* per_sample,
                 float* dscore, int B, int E, double sigma_min, double sigma_max, int likelihood_weighting, int reduce_mean,
                 void* stream) {
    if (!score || !perturbed || !batch || !t || !per_sample) return fail("null argument");
    hipLaunchKernelGGL(sm_loss_kernel, dim3((unsigned)B), dim3(64), 0, (hipStream_t)stream, score, perturbed, batch, t, per_sample, dscore, B, E,
                       (float)sigma_min, (float)(sigma_max / sigma_min), g_const(sigma_min, sigma_max), likelihood_weighting, reduce_mean,
                       20, 10, 1e-2f);
    HIP_OK(hipGetLastError());
    return 0;
}

int rdmi_em_update(const float* x, const float* score, const float* z, const float* t, float* x_out, float* x_mean_out,
                   int B, int E, int N, double sigma_min, double sigma_max, void* stream) {
    if (!x || !score || !z || !t || !x_out) return fail("null argument");
    hipLaunchKernelGGL(em_update_kernel, dim3((unsigned)ceil_div(B * E, RDMI_THREADS)), dim3(RDMI_THREADS), 0,
                       (hipStream_t)stream, x, score, z, t, (const float*)nullptr, (const StepState*)nullptr, x_out,
                       x_mean_out, (float*)nullptr, B, E, N, (float)sigma_min, (float)(sigma_max / sigma_min),
                       g_const(sigma_min, sigma_max), 0);
    HIP_OK(hipGetLastError());
    return 0;
}

int rdmi_langevin_update(const float* x, const float* score, const float* z, float* x_out, float* x_mean_out,
                         float* scratch, int B, int E, float snr, void* stream) {
    if (!x || !score || !z || !x_out || !scratch) return fail("null argument");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(row_norms_kernel, dim3((unsigned)B), dim3(64), 0, s, score, z, (const StepState*)nullptr, scratch, B, E, 0);
    hipLaunchKernelGGL(langevin_update_kernel, dim3((unsigned)ceil_div(B * E, RDMI_THREADS)), dim3(RDMI_THREADS), 64, s, x,
                       score, z, (const float*)scratch, (const StepState*)nullptr, x_out, x_mean_out, B, E, snr, 0);
    HIP_OK(hipGetLastError());
    return 0;
}

int rdmi_pc_sample(rdmi_ctx* c, float* x, const float* labels, const float* weight, const float* noise, float* trace,
                   const float* teacher, int B, const rdmi_pc_opts* o, unsigned flags, void* stream) {
    if (!c || !x || !o) return fail("null argument");
    if (o->N < 2) return fail("N=%d: need at least 2 scales", o->N);
    if (o->corrector != 0 && o->corrector != 1) return fail("unknown corrector id %d", o->corrector);
    const int NBm = o->use_cfg ? 2 * B : B;
    if (NBm > c->max_batch) return fail("model batch %d > max_batch=%d", NBm, c->max_batch);
    if (o->use_cfg && !labels) return fail("use_cfg=1 needs class_labels");
    hipStream_t s = (hipStream_t)stream;
    if (int e = maybe_repack(c, flags, s)) return e;
    const int E = c->H * c->W * c->arch.channels;
    const long BE = (long)B * E;
    const float smin = (float)o->sigma_min, ratio = (float)(o->sigma_max / o->sigma_min);
    const float gc = g_const(o->sigma_min, o->sigma_max);
    const unsigned gBE = (unsigned)ceil_div((int)BE, RDMI_THREADS);
    const uint64_t eoff = (uint64_t)o->seq_offset * (uint64_t)E;

    // timesteps = torch.linspace(T=1, eps, N): fp32 step, fma(step, i, start) / fma(-step, N-1-i, end)
    std::vector<float> ts((size_t)o->N);
    {
        const float start = 1.0f, end = o->eps;
        const double step = (double)(float)((end - start) / (float)(o->N - 1));
        for (int i = 0; i < o->N; ++i)
            ts[(size_t)i] = (float)(i < o->N / 2 ? (double)start + step * i : (double)end - step * (o->N - 1 - i));
    }
    // The time path of the embedding (Fourier -> time_mlp.0 -> SiLU -> time_mlp.2, + both biases) depends only on the
    // update's t, which is known up front: evaluate it for ALL updates in two launches (one row per update) instead of two
    // launches per score evaluation; per evaluation only the batched Dense_0 GEMM remains, which adds the label embedding
    // to the shared row in its prologue.
    const int nupd = o->N - 1, T = c->temb, Np_t = (T + 63) & ~63, Np_d = (c->dense_total + 63) & ~63;
    if (c->tt_cap < nupd) {
        if (c->d_tt) { HIP_OK(hipFree(c->d_tt)); HIP_OK(hipFree(c->d_th1)); HIP_OK(hipFree(c->d_ts)); c->d_tt = c->d_th1 = c->d_ts = nullptr; c->tt_cap = 0; }
        HIP_OK(hipMalloc((void**)&c->d_tt, (size_t)pad16(nupd) * T * sizeof(float)));
        HIP_OK(hipMalloc((void**)&c->d_th1, (size_t)pad16(nupd) * T * sizeof(float)));
        HIP_OK(hipMalloc((void**)&c->d_ts, (size_t)pad16(nupd) * sizeof(float)));
        c->tt_cap = nupd;
    }
    const rdmi_arch& a = c->arch;
    {
        // the time grid is formed on the device by the same double-precision expression as the host copy above (no host
        // buffer has to outlive this call: every rdmi entry point is asynchronous on `stream`)
        hipLaunchKernelGGL(linspace_kernel, dim3((unsigned)ceil_div(nupd, RDMI_THREADS)), dim3(RDMI_THREADS), 0, s, c->d_ts, nupd, o->N, 1.0f, o->eps);
        LinArgs l{};
        l.M = nupd; l.fourW = P(c, "time_embed.W"); l.nfour = a.nf; l.X = c->d_ts; l.x_mod = nupd; l.t_is_time = 1; l.smin = smin; l.ratio = ratio;
        l.pre = 2; l.K = pad16(2 * a.nf); l.W = c->d_w + c->w_t0; l.Npad = Np_t; l.N = T; l.bias = P(c, "time_mlp.0.bias"); l.Y = c->d_th1; l.ldy = T;
        LinArgs l2{};
        l2.M = nupd; l2.X = c->d_th1; l2.ldx = T; l2.pre = 1; l2.K = T; l2.W = c->d_w + c->w_t2; l2.Npad = Np_t; l2.N = T;
        l2.bias = P(c, "time_mlp.2.bias"); l2.Y = c->d_tt; l2.ldy = T;
        if (a.conditional) l2.bl = P(c, "label_emb.bias");
        ProfScope ps(c, s, "linear_mfma(time path, all updates)", 2.0 * nupd * T * (2 * a.nf + T));
        hipLaunchKernelGGL(linear_mfma_kernel<1>, dim3((unsigned)ceil_div(nupd, 16), (unsigned)(Np_t / 64)), dim3(RDMI_THREADS), 0, s, l);
        hipLaunchKernelGGL(linear_mfma_kernel<1>, dim3((unsigned)ceil_div(nupd, 16), (unsigned)(Np_t / 64)), dim3(RDMI_THREADS), 0, s, l2);
        HIP_OK(hipGetLastError());
    }
    // Dense_0 (the 17 per-block projections of SiLU(temb), RD/models/layerspp.py:202) depends on (update, sample) only through
    // the update's time row and the sample's label: it is evaluated for a whole CHUNK of updates in one GEMM launch
    // ([U * NBm] rows: a real dense contraction instead of 999 launches of 256 rows) into a buffer of at most ~256 MB;
    // row (u, n) = Dense_0(SiLU(tt[u] + label_emb(labels[n]))): the same sums in the same order as the stand-alone path.
    const size_t dense_row_floats = (size_t)NBm * c->dense_total;
    const int U = (int)std::max<size_t>(1, std::min<size_t>((size_t)nupd, ((size_t)256 << 20) / (dense_row_floats * sizeof(float))));
    if (c->dense_all_cap < (size_t)U * dense_row_floats) {
        if (c->d_dense_all) { HIP_OK(hipFree(c->d_dense_all)); c->d_dense_all = nullptr; c->dense_all_cap = 0; }
        HIP_OK(hipMalloc((void**)&c->d_dense_all, (size_t)U * dense_row_floats * sizeof(float)));
        c->dense_all_cap = (size_t)U * dense_row_floats;
    }
    auto dense_chunk = [&](int i0) -> int {
        const int u = std::min(U, nupd - i0);
        LinArgs l3{};
        l3.M = u * NBm; l3.X = c->d_tt + (size_t)i0 * T; l3.ldx = T; l3.pre = 3; l3.K = T; l3.row_div = NBm;
        if (a.conditional && labels) { l3.labels = labels; l3.Wl = P(c, "label_emb.weight"); l3.ncls = a.num_classes; l3.label_rows = B; }
        l3.W = c->d_w + c->w_dense; l3.Npad = Np_d; l3.N = c->dense_total;
        l3.bias = c->d_w + c->b_dense; l3.Y = c->d_dense_all; l3.ldy = c->dense_total;
        ProfScope ps(c, s, "linear_mfma(Dense_0 x all, chunk of updates)", 2.0 * l3.M * T * c->dense_total);
        hipLaunchKernelGGL((linear_mfma_kernel<1, 16>), dim3((unsigned)ceil_div(l3.M, 16), (unsigned)(Np_d / 64)), dim3(RDMI_THREADS), 0, s, l3);
        HIP_OK(hipGetLastError());
        return 0;
    };
    // one score evaluation at update i's shared time: the raw network output lands in d_s2 ([2B] with CFG, [B] without)
    auto net_eval = [&](int i, float t) -> int {
        FwdIn f{x, o->use_cfg ? B : 0, nullptr, 0, t, 1, smin, ratio, labels, B, c->d_s2, NBm};
        f.tt_row = c->d_tt + (size_t)i * T;
        f.dense_rows = c->d_dense_all + (size_t)(i % U) * dense_row_floats;
        return run_forward(c, f, s);
    };
    uint32_t draw = 0;
    for (int i = 0; i < o->N - 1; ++i) {
        const float t = ts[(size_t)i];
        if (i % U == 0) { if (int e = dense_chunk(i)) return e; }
        if (o->corrector == 1) {
            for (int k = 0; k < o->n_steps_each; ++k, ++draw) {
                if (int e = net_eval(i, t)) return e;
                ProfScope ps(c, s, "langevin_update", 0);
                hipLaunchKernelGGL(langevin_prep_kernel, dim3((unsigned)B), dim3(64), 0, s, (const float*)c->d_s2, weight,
                                   noise ? noise + (size_t)draw * BE : (const float*)nullptr, c->d_score, c->d_z, c->d_norms, B, E,
                                   o->use_cfg, (uint64_t)o->seed, eoff, draw);
                hipLaunchKernelGGL(langevin_update_kernel, dim3(gBE), dim3(RDMI_THREADS), 64, s, (const float*)x,
                                   (const float*)c->d_score, (const float*)c->d_z, (const float*)c->d_norms, (const StepState*)nullptr, x,
                                   (float*)nullptr, B, E, o->snr, 0);
            }
        }
        if (int e = net_eval(i, t)) return e;
        {
            ProfScope ps(c, s, "em_fused", 0);
            hipLaunchKernelGGL(em_fused_kernel, dim3(gBE), dim3(RDMI_THREADS), 0, s, (const float*)x, (const float*)c->d_s2, weight,
                               noise ? noise + (size_t)draw * BE : (const float*)nullptr, x, trace ? trace + (size_t)i * BE : (float*)nullptr,
                               teacher ? teacher + (size_t)i * BE : (const float*)nullptr, B, E, o->N, t, smin, ratio, gc, o->use_cfg,
                               (uint64_t)o->seed, eoff, draw);
            ++draw;
        }
        HIP_OK(hipGetLastError());
        if (c->profiling && (i % 16 == 15)) prof_collect(c);
    }
    if (c->profiling) prof_collect(c);
    return 0;
}

// ---- probability-flow ODE sampler: scipy's RK45 (Dormand-Prince 5(4), scipy/integrate/_ivp/rk.py) with the state on the device ----
int rdmi_ode_sample(rdmi_ctx* c, float* x, const float* labels, const float* weight, int B, const rdmi_ode_opts* o, int* nfev_out,
                    double* t_final, unsigned flags, void* stream) {
    if (!c || !x || !o) return fail("null argument");
    const int NBm = o->use_cfg ? 2 * B : B;
    if (NBm > c->max_batch) return fail("model batch %d > max_batch=%d", NBm, c->max_batch);
    if (o->use_cfg && !labels) return fail("use_cfg=1 needs class_labels");
    if (!(o->rtol > 0) || !(o->atol > 0)) return fail("rtol / atol must be positive");
    hipStream_t s = (hipStream_t)stream;
    if (int e = maybe_repack(c, flags, s)) return e;
    const int E = c->H * c->W * c->arch.channels;
    const long n = (long)B * E;
    const unsigned gn = (unsigned)ceil_div((int)n, RDMI_THREADS);
    const float smin = (float)o->sigma_min, ratio = (float)(o->sigma_max / o->sigma_min), gc = g_const(o->sigma_min, o->sigma_max);
    // Dormand-Prince tableau as scipy holds it (rk.py: class RK45)
    static const double Cc[6] = {0, 1.0 / 5, 3.0 / 10, 4.0 / 5, 8.0 / 9, 1};
    static const double A[6][5] = {{0, 0, 0, 0, 0}, {1.0 / 5, 0, 0, 0, 0}, {3.0 / 40, 9.0 / 40, 0, 0, 0}, {44.0 / 45, -56.0 / 15, 32.0 / 9, 0, 0},
                                   {19372.0 / 6561, -25360.0 / 2187, 64448.0 / 6561, -212.0 / 729, 0},
                                   {9017.0 / 3168, -355.0 / 33, 46732.0 / 5247, 49.0 / 176, -5103.0 / 18656}};
    static const double Bc[6] = {35.0 / 384, 0, 500.0 / 1113, 125.0 / 192, -2187.0 / 6784, 11.0 / 84};
    static const double Ec[7] = {-71.0 / 57600, 0, 71.0 / 16695, -71.0 / 1920, 17253.0 / 339200, -22.0 / 525, 1.0 / 40};
    const double SAFETY = 0.9, MIN_FACTOR = 0.2, MAX_FACTOR = 10, err_exp = -1.0 / 5;
    double *d_y = nullptr, *d_ynew = nullptr, *d_K = nullptr, *d_f1 = nullptr, *d_red = nullptr;
    float* d_x32 = nullptr;
    auto cleanup = [&]() { for (void* p : {(void*)d_y, (void*)d_ynew, (void*)d_K, (void*)d_f1, (void*)d_red, (void*)d_x32}) if (p) (void)hipFree(p); };
#define ODE_OK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { cleanup(); return fail("%s failed: %s", #expr, hipGetErrorString(e_)); } } while (0)
    ODE_OK(hipMalloc((void**)&d_y, n * sizeof(double)));
    ODE_OK(hipMalloc((void**)&d_ynew, n * sizeof(double)));
    ODE_OK(hipMalloc((void**)&d_K, 7 * n * sizeof(double)));
    ODE_OK(hipMalloc((void**)&d_f1, n * sizeof(double)));
    ODE_OK(hipMalloc((void**)&d_red, 4 * sizeof(double)));
    ODE_OK(hipMalloc((void**)&d_x32, n * sizeof(float)));
    int nfev = 0;
    // fun(t, x32) -> Kout (float64 copy of the float32 drift * bump)
    auto fun = [&](double t, double* Kout) -> int {
        FwdIn f{d_x32, o->use_cfg ? B : 0, nullptr, 0, (float)t, 1, smin, ratio, labels, B, c->d_s2, NBm};     // vec_t = ones(B) * t  (float32)
        if (int e = run_forward(c, f, s)) return e;
        hipLaunchKernelGGL(ode_rhs_kernel, dim3(gn), dim3(RDMI_THREADS), 0, s, (const float*)c->d_s2, weight, (const float*)d_x32, Kout, B, E, (float)t, smin, ratio, gc,
                           o->use_cfg, o->moll);
        ++nfev;
        return 0;
    };
    auto norm_of = [&](int mode, const double* a, const double* b, const double* y0, double h, double* out_host) -> int {
        OdeNormArgs q{a, b, y0, d_K, n, mode, o->atol, o->rtol, h, {Ec[0], Ec[1], Ec[2], Ec[3], Ec[4], Ec[5], Ec[6]}, d_red, 0};
        hipLaunchKernelGGL(ode_norm_kernel, dim3(1), dim3(RDMI_THREADS), RDMI_THREADS * sizeof(double), s, q);
        double ss = 0;
        if (hipMemcpyAsync(&ss, d_red, sizeof(double), hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) return fail("ode: read-back failed");
        *out_host = std::sqrt(ss) / std::sqrt((double)n);            // scipy: norm(x) = ||x|| / sqrt(x.size)
        return 0;
    };
    const double t0 = o->T, t_bound = o->eps, direction = t_bound >= t0 ? 1.0 : -1.0;
    double t = t0;
    int rc = 0;
    hipLaunchKernelGGL(ode_f2d_kernel, dim3(gn), dim3(RDMI_THREADS), 0, s, (const float*)x, d_y, n);
    hipLaunchKernelGGL(ode_stage_kernel, dim3(gn), dim3(RDMI_THREADS), 0, s, (const double*)d_y, (const double*)d_K, n, 0, 0., 0., 0., 0., 0., 0., 0., (double*)nullptr, d_x32);
    if ((rc = fun(t, d_K))) { cleanup(); return rc; }                // self.f = fun(t0, y0)
    double h_abs;
    if (o->first_step > 0) h_abs = o->first_step;
    else {   // select_initial_step (scipy/integrate/_ivp/common.py), order = 4
        const double interval = std::fabs(t_bound - t0);
        double d0 = 0, d1 = 0, d2 = 0;
        if ((rc = norm_of(0, d_y, nullptr, d_y, 0, &d0)) || (rc = norm_of(0, d_K, nullptr, d_y, 0, &d1))) { cleanup(); return rc; }
        double h0 = (d0 < 1e-5 || d1 < 1e-5) ? 1e-6 : 0.01 * d0 / d1;
        h0 = std::min(h0, interval);
        hipLaunchKernelGGL(ode_axpy_kernel, dim3(gn), dim3(RDMI_THREADS), 0, s, (const double*)d_y, (const double*)d_K, h0 * direction, n, d_x32);
        if ((rc = fun(t0 + h0 * direction, d_f1))) { cleanup(); return rc; }
        if ((rc = norm_of(1, d_f1, d_K, d_y, 0, &d2))) { cleanup(); return rc; }
        d2 /= h0;
        const double h1 = (d1 <= 1e-15 && d2 <= 1e-15) ? std::max(1e-6, h0 * 1e-3) : std::pow(0.01 / std::max(d1, d2), 1.0 / 5);
        h_abs = std::min(std::min(100 * h0, h1), interval);
    }
    int steps = 0;
    while (direction * (t - t_bound) < 0) {
        if (o->max_steps > 0 && steps >= o->max_steps) break;
        const double min_step = 10 * std::fabs(std::nextafter(t, direction * INFINITY) - t);
        if (h_abs < min_step) h_abs = min_step;
        bool accepted = false, rejected = false;
        double t_new = t, h = 0;
        while (!accepted) {
            if (h_abs < min_step) { cleanup(); return fail("ode: required step size is less than spacing between numbers (scipy: TOO_SMALL_STEP) at t=%g", t); }
            h = h_abs * direction;
            t_new = t + h;
            if (direction * (t_new - t_bound) > 0) t_new = t_bound;
            h = t_new - t;
            h_abs = std::fabs(h);
            for (int st = 1; st < 6; ++st) {                         // rk_step: K[s] = fun(t + c_s h, y + h * K[:s].T a_s[:s])
                hipLaunchKernelGGL(ode_stage_kernel, dim3(gn), dim3(RDMI_THREADS), 0, s, (const double*)d_y, (const double*)d_K, n, st, A[st][0], A[st][1], A[st][2], A[st][3],
                                   A[st][4], 0., h, (double*)nullptr, d_x32);
                if ((rc = fun(t + Cc[st] * h, d_K + (size_t)st * n))) { cleanup(); return rc; }
            }
            hipLaunchKernelGGL(ode_stage_kernel, dim3(gn), dim3(RDMI_THREADS), 0, s, (const double*)d_y, (const double*)d_K, n, 6, Bc[0], Bc[1], Bc[2], Bc[3], Bc[4], Bc[5], h,
                               d_ynew, d_x32);                       // y_new = y + h * K[:-1].T B
            if ((rc = fun(t + h, d_K + (size_t)6 * n))) { cleanup(); return rc; }      // f_new -> K[-1]
            double err = 0;
            if ((rc = norm_of(2, nullptr, d_ynew, d_y, h, &err))) { cleanup(); return rc; }
            if (err < 1) {
                double factor = err == 0 ? MAX_FACTOR : std::min(MAX_FACTOR, SAFETY * std::pow(err, err_exp));
                if (rejected) factor = std::min(1.0, factor);
                h_abs *= factor;
                accepted = true;
            } else {
                h_abs *= std::max(MIN_FACTOR, SAFETY * std::pow(err, err_exp));
                rejected = true;
            }
        }
        // accept: y <- y_new, f <- f_new (K[0] <- K[6])
        std::swap(d_y, d_ynew);
        ODE_OK(hipMemcpyAsync(d_K, d_K + (size_t)6 * n, n * sizeof(double), hipMemcpyDeviceToDevice, s));
        t = t_new;
        ++steps;
    }
    hipLaunchKernelGGL(ode_d2f_kernel, dim3(gn), dim3(RDMI_THREADS), 0, s, (const double*)d_y, x, n);
    ODE_OK(hipStreamSynchronize(s));
    if (nfev_out) *nfev_out = nfev;
    if (t_final) *t_final = t;
    if (o->h_next_out) *o->h_next_out = h_abs;
    cleanup();
#undef ODE_OK
    if (c->profiling) prof_collect(c);
    return 0;
}

// ---- multi-tensor optimizer (clip + Adam/AdamW + EMA), csrc/opt_kernels.h ---------------------------------------------
struct rdmi_opt {
    std::vector<OptSlot> h_slots;           // host mirror of the slot table (rdmi_opt_update_slots uploads from here)
    int nslots = 0, nchunks = 0;
    OptSlot* d_slots = nullptr; OptChunk* d_chunks = nullptr; int* d_first = nullptr; float* d_partial = nullptr; float* d_norm = nullptr;
};

int rdmi_opt_create(const rdmi_opt_slot* slots, int n, rdmi_opt** out) {
    if (!slots || !out || n < 1) return fail("rdmi_opt_create: null / empty slot table");
    static_assert(sizeof(rdmi_opt_slot) == sizeof(OptSlot), "ABI slot layout");
    std::vector<OptChunk> chunks;
    std::vector<int> first((size_t)n + 1);
    for (int t = 0; t < n; ++t) {
        if (!slots[t].param || !slots[t].grad || !slots[t].exp_avg || !slots[t].exp_avg_sq) return fail("rdmi_opt_create: slot %d has a null pointer", t);
        if (slots[t].numel >= (1ull << 32)) return fail("rdmi_opt_create: slot %d has %llu elements (>= 2^32)", t, slots[t].numel);
        first[(size_t)t] = (int)chunks.size();
        for (unsigned long long o = 0; o < slots[t].numel; o += OPT_CHUNK) chunks.push_back({t, (unsigned)o});
    }
    first[(size_t)n] = (int)chunks.size();
    rdmi_opt* q = new rdmi_opt();
    q->nslots = n; q->nchunks = (int)chunks.size();
    auto bail = [&](const char* what) { rdmi_opt_destroy(q); return fail("rdmi_opt_create: %s failed", what); };
    if (hipMalloc((void**)&q->d_slots, (size_t)n * sizeof(OptSlot)) != hipSuccess) return bail("hipMalloc");
    if (hipMalloc((void**)&q->d_chunks, std::max<size_t>(chunks.size(), 1) * sizeof(OptChunk)) != hipSuccess) return bail("hipMalloc");
    if (hipMalloc((void**)&q->d_first, ((size_t)n + 1) * sizeof(int)) != hipSuccess) return bail("hipMalloc");
    if (hipMalloc((void**)&q->d_partial, std::max<size_t>(chunks.size(), 1) * sizeof(float)) != hipSuccess) return bail("hipMalloc");
    if (hipMalloc((void**)&q->d_norm, 2 * sizeof(float)) != hipSuccess) return bail("hipMalloc");
    if (hipMemcpy(q->d_slots, slots, (size_t)n * sizeof(OptSlot), hipMemcpyHostToDevice) != hipSuccess) return bail("hipMemcpy");
    if (hipMemcpy(q->d_chunks, chunks.data(), chunks.size() * sizeof(OptChunk), hipMemcpyHostToDevice) != hipSuccess) return bail("hipMemcpy");
    if (hipMemcpy(q->d_first, first.data(), first.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) return bail("hipMemcpy");
    q->h_slots.assign(reinterpret_cast<const OptSlot*>(slots), reinterpret_cast<const OptSlot*>(slots) + n);
    *out = q;
    return 0;
}

// New pointers for the same tensors (same count, same element counts): e.g. the gradients of a step landed in a fresh buffer.
// One asynchronous upload on `stream`; no allocation, no synchronisation.
int rdmi_opt_update_slots(rdmi_opt* q, const rdmi_opt_slot* slots, int n, void* stream) {
    if (!q || !slots) return fail("rdmi_opt_update_slots: null argument");
    if (n != q->nslots) return fail("rdmi_opt_update_slots: %d slots, the table was created with %d", n, q->nslots);
    const OptSlot* ns = reinterpret_cast<const OptSlot*>(slots);
    for (int t = 0; t < n; ++t) {
        if (ns[t].n != q->h_slots[(size_t)t].n) return fail("rdmi_opt_update_slots: slot %d changed its element count", t);
        if (!slots[t].param || !slots[t].grad || !slots[t].exp_avg || !slots[t].exp_avg_sq) return fail("rdmi_opt_update_slots: slot %d has a null pointer", t);
        if ((q->h_slots[(size_t)t].ema == nullptr) != (ns[t].ema == nullptr)) return fail("rdmi_opt_update_slots: slot %d gained or lost its EMA shadow", t);
    }
    q->h_slots.assign(ns, ns + n);
    HIP_OK(hipMemcpyAsync(q->d_slots, q->h_slots.data(), (size_t)n * sizeof(OptSlot), hipMemcpyHostToDevice, (hipStream_t)stream));
    return 0;
}

int rdmi_opt_destroy(rdmi_opt* q) {
    if (!q) return 0;
    for (void* p : {(void*)q->d_slots, (void*)q->d_chunks, (void*)q->d_first, (void*)q->d_partial, (void*)q->d_norm}) if (p) (void)hipFree(p);
    delete q;
    return 0;
}

int rdmi_opt_step(rdmi_opt* q, const rdmi_opt_hyper* hy, float* total_norm_out, void* stream) {
    if (!q || !hy) return fail("rdmi_opt_step: null argument");
    if (hy->step < 1) return fail("rdmi_opt_step: step=%d (the first update is step 1)", hy->step);
    hipStream_t s = (hipStream_t)stream;
    OptHyper h{};
    h.lr = hy->lr; h.beta1 = hy->beta1; h.beta2 = hy->beta2; h.eps = hy->eps; h.weight_decay = hy->weight_decay; h.decoupled_wd = hy->decoupled_wd;
    // python-float arithmetic of torch.optim.adam._single_tensor_adam, rounded to fp32 where the kernel consumes it
    const double bc1 = 1.0 - std::pow((double)hy->beta1_d, (double)hy->step), bc2 = 1.0 - std::pow((double)hy->beta2_d, (double)hy->step);
    h.step_size = (float)((double)hy->lr_d / bc1);
    h.bc2_sqrt = (float)std::sqrt(bc2);
    h.one_minus_beta1 = (float)(1.0 - hy->beta1_d); h.one_minus_beta2 = (float)(1.0 - hy->beta2_d);
    h.max_norm = hy->max_norm;
    h.one_minus_ema_decay = (float)(1.0 - hy->ema_decay_d);
    h.write_back_grad = hy->write_back_grad;
    hipLaunchKernelGGL(opt_sumsq_kernel, dim3((unsigned)q->nchunks), dim3(RDMI_THREADS), 16, s, (const OptSlot*)q->d_slots, (const OptChunk*)q->d_chunks, q->d_partial);
    hipLaunchKernelGGL(opt_norm_kernel, dim3(1), dim3(RDMI_THREADS), RDMI_THREADS * sizeof(float), s, (const float*)q->d_partial, (const int*)q->d_first, q->nslots,
                       hy->max_norm, q->d_norm);
    hipLaunchKernelGGL(opt_adam_ema_kernel, dim3((unsigned)q->nchunks), dim3(RDMI_THREADS), 0, s, (const OptSlot*)q->d_slots, (const OptChunk*)q->d_chunks, h,
                       (const float*)q->d_norm);
    HIP_OK(hipGetLastError());
    if (total_norm_out) HIP_OK(hipMemcpyAsync(total_norm_out, q->d_norm, sizeof(float), hipMemcpyDeviceToDevice, s));
    return 0;
}

int rdmi_get_tap(rdmi_ctx* c, const char* name, float* dst, size_t dst_numel, int* C, int* H, int* W, void* stream) {
    if (!c || !name || !dst) return fail("null argument");
    hipStream_t s = (hipStream_t)stream;
    const std::string nm(name);
    if (c->tiled) {      // the tiled plan never reuses a tensor: the output of a block is its last conv (Conv_1 / NIN_3)
        for (auto it = c->tl.rbegin(); it != c->tl.rend(); ++it) {
            if (it->kind != 0 || !(it->name == nm || it->name == nm + ".Conv_1" || it->name == nm + ".NIN_3")) continue;
            const TConvArgs& a = it->conv;
            if (C) *C = a.Cout; if (H) *H = a.Ho; if (W) *W = a.Wo;
            const size_t per = (size_t)a.Cout * a.Ho * a.Wo;
            const int nb = (int)std::min<size_t>(dst_numel / per, (size_t)c->max_batch);
            hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3((unsigned)ceil_div((int)(nb * per), RDMI_THREADS)), dim3(RDMI_THREADS), 0, s, (const float*)a.out, dst, nb,
                               a.Ho * a.Wo, a.Cout, 0);
            HIP_OK(hipGetLastError());
            return 0;
        }
        return fail("no activation named '%s' in the tiled plan", name);
    }
    if (!c->debug_taps) return fail("taps need RDMI_DEBUG_TAPS=1 at rdmi_create (buffers are reused otherwise)");
    if (nm == "temb") {
        if (C) *C = c->temb; if (H) *H = 1; if (W) *W = 1;
        HIP_OK(hipMemcpyAsync(dst, c->d_temb, std::min(dst_numel, (size_t)c->max_batch * c->temb) * sizeof(float), hipMemcpyDeviceToDevice, s));
        return 0;
    }
    for (auto& t : c->tensors) {
        if (t.name != nm) continue;
        if (C) *C = t.C; if (H) *H = t.H; if (W) *W = t.W;
        const size_t per = t.per_sample();
        const int nb = (int)std::min<size_t>(dst_numel / per, (size_t)c->max_batch);
        hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3((unsigned)ceil_div((int)(nb * per), RDMI_THREADS)), dim3(RDMI_THREADS), 0, s,
                           (const float*)(c->ws + t.off * (size_t)c->max_batch), dst, nb, t.H * t.W, t.C, c->arch.compute_dtype == 1 ? 1 : 0);
        HIP_OK(hipGetLastError());
        return 0;
    }
    return fail("no activation named '%s'", name);
}

int rdmi_set_profiling(rdmi_ctx* c, int enabled) {
    if (!c) return fail("null context");
    c->profiling = enabled != 0;
    c->prof.clear();
    c->ev_used = 0;
    return 0;
}

int rdmi_get_profile(rdmi_ctx* c, int index, const char** kernel_name, double* total_ms, long* launches, double* flops) {
    if (!c || index < 0 || index >= (int)c->prof.size()) return 1;
    const ProfEntry& e = c->prof[(size_t)index];
    if (kernel_name) *kernel_name = e.name.c_str();
    if (total_ms) *total_ms = e.ms;
    if (launches) *launches = e.launches;
    if (flops) *flops = e.flops;
    return 0;
}

}  // extern "C"
