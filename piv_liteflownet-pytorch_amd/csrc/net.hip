// Network object behind pivlfn_create / pivlfn_forward: weight repacking, workspace plan and the
// coarse-to-fine level pipeline of LiteFlowNet.forward (/root/reference/src/models.py:319-370) expressed as a
// fixed sequence of gfx950 kernel launches on one stream.  No allocation, no host sync inside forward.
#include <algorithm>
#include <map>
#include <string>
#include <vector>
#include "common.h"

namespace pivlfn {

static const int K_LEVEL[7] = {0, 7, 7, 5, 5, 3, 3};            // src/models.py:161,205,225
static const int C_FEAT[7] = {0, 32, 32, 64, 96, 128, 192};     // src/models.py:70-106
static const int C_MATCH[7] = {0, 64, 64, 64, 96, 128, 192};    // NetC_ext: src/models.py:124,353-357

static inline int rup(int a, int b) { return (a + b - 1) / b * b; }

// Split-K scratch (conv_mfma.hip): a layer is split only when one image has <= 128 workgroups of 128 px x 32 channels, i.e. at
// most 128*128*32 partial sums per share and image, and into at most 8 shares.
static const size_t KSPLIT_FLOATS = (size_t)8 * 128 * 128 * 32;
// The reduction of the shares is a kernel of its own (16 launches of 6.7 us at 1024^2).  Round 6 let the share that arrives last at a
// tile do it (arrival counter, agent-scope release / acquire fences; same order, same bits): the forward got 0.42 ms SLOWER
// (profiles/r06_item5_net_ab.log) -- a release at agent scope writes back the XCD's whole L2 (eight L2s that are not coherent with
// each other), ~500 workgroups x 16 layers of it while the side stream keeps 1.4 GB of dirty lines going.  Not adopted.

void pack_conv_h(const float *w, int cout, int cin, int taps, const int *creal, const int *cload, const int *coff, int nseg,
                 std::vector<unsigned short> &pk, int *nchunk_out);      // conv_f16.hip
void pack_conv_x(const float *w, int cout, int cin, int taps, const int *creal, const int *cload, const int *coff, int nseg,
                 std::vector<unsigned short> &pk, int *nchunk_out, float *out_scale);      // conv_split.hip
void pack_conv_x_tail(const float *w, int cout, int cin, int c_first, int c_real, float scale_inv, std::vector<unsigned short> &pk);
void pack_conv_w(const float *w, int cout, int cin, const int *creal, const int *cload, const int *coff, int nseg,
                 std::vector<float> &pk, int *nchunk_out);
#ifdef PIVLFN_TOOLS
void pack_conv_w4(const float *w, int cout, int cin, const int *creal, const int *cload, const int *coff, int nseg,
                 std::vector<float> &pk, int *nchunk_out);      // tools/kernels/conv_wino4.hip
#endif
void pack_conv_wb(const float *w, int cout, int cin, const int *creal, const int *cload, const int *coff, int nseg,
                  std::vector<unsigned short> &pk, int *nstep_out);      // conv_wino_b3.hip

struct ConvW {
    float *wpk = nullptr, *bias = nullptr;
    int cout = 0, cout_pad = 0, KH = 0, KW = 0, nchunk = 0, tail = 0, cin = 0;
    void *wpk_h = nullptr;         // fp16 packing for conv_f16.hip (K chunks of 16 channels)
    int nchunk_h = 0;
    void *wpk_x = nullptr;         // three-piece fp16 packing for conv_split.hip (fp32 by exact splitting); nullptr = unsupported geometry
    int nchunk_x = 0;
    float scale_x = 1.f;           // 2^-k undoing the weight scale of wpk_x
    void *wtail_x = nullptr;       // 3 x 3 layers whose staged channels end in a 4-lane tail: that chunk with taps folded into K
    float *wpk_c = nullptr;        // (7 x 1) layers from 32 channels: A fragments of conv_col7_kernel, [4][7][2][64][4]
    float *wpk_r = nullptr, *wpk_r12 = nullptr;   // the (1 x 7) 49 -> 49 layer: A fragments of conv_row7_kernel, [4][7][3][64][4] and [4][7][64]
    float *wpk_w = nullptr;        // 3 x 3 layers: Winograd-domain weights G g G^T in fragment order (conv_wino.hip)
    float *wpk_w4 = nullptr;       // the same for F(4x4, 3x3): 36 planes (conv_wino4.hip)
    int nchunk_w = 0, nchunk_w4 = 0;
    void *wpk_wb = nullptr;        // 3 x 3 layers with whole 64-channel groups: the Winograd-domain weights as three bf16 pieces each (conv_wino_b3.hip)
    int nstep_wb = 0;
};

struct LevelW {
    float *upconv = nullptr, *upcorr = nullptr;    // depthwise k4 weights [16 taps][C4]
    ConvW M[6], S[6], R[6], feat, dist0, dist1;   // M/S: nstack hidden 3x3 layers, then the k x k head at index nstack
    float *headM = nullptr, *headS = nullptr;      // VALU flow-head weights [k*k][8][4][2]
    float hbM[2] = {0.f, 0.f}, hbS[2] = {0.f, 0.f};
    float *wx = nullptr, *wy = nullptr;
    float bx = 0.f, by = 0.f;
};

}  // namespace pivlfn

struct pivlfn_net;
struct pivlfn_conv {
    pivlfn::ConvW cw;
    int cin = 0;
    int nsrc = 1;              // sources of the layer (pivlfn_conv_create_cat: a convolution over a channel concatenation)
    int src_real[3] = {0, 0, 0};
    float *scratch = nullptr;  // split-K scratch (KSPLIT_FLOATS), allocated by conv_create
    float *head = nullptr;     // set when the layer is a 32->2 kxk flow head
    float hb[2] = {0.f, 0.f};
    pivlfn_net *owner = nullptr;   // holds the device allocations
};

struct pivlfn_net {
    float scale[7];
    int lowest;
    int nstack = 3;                // hidden conv_M / conv_S layers: 3 = LiteFlowNet (src/models.py:154-163), 5 = LiteFlowNet2 (:487-500)
    int width[5] = {128, 64, 32, 0, 0};
    float mean[6];
    int precision = 0;             // PIVLFN_PRECISION_*: 0 fp32 instruction (default; 3x3 stride-1 layers by Winograd), 1 fp16 multiplicands,
                                   // 2 / 3 fp32 by operand splitting, 4 fp32 instruction with direct convolution everywhere
    pivlfn::ConvW netc[10];
    pivlfn::ConvW ext[3];          // index by level (1,2)
    pivlfn::LevelW lv[7];
    std::vector<void *> allocs;
    // side stream for the flow-independent 1x1 convs (NetC_ext, moduleFeat): they overlap the latency-bound coarse levels
    hipStream_t side = nullptr;
    float *fuse1_w = nullptr, *fuse1_b = nullptr;      // level 1: NetC_ext + moduleFeat as 1 x 1 layers inside NetC.conv1's kernel (Conv1Fuse)
    hipEvent_t ev_fork[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}, ev_join[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    // measurement hooks
    int prof_level = 0;
    std::vector<hipEvent_t> ev;
    size_t ev_used = 0;
    long ev_dropped = 0;
};

namespace pivlfn {

typedef std::map<std::string, const pivlfn_tensor *> TMap;

static int upload(pivlfn_net *net, const std::vector<float> &h, float **dev)
{
    void *d = nullptr;
    PIV_CHECK_HIP(hipMalloc(&d, h.size() * sizeof(float)));
    net->allocs.push_back(d);
    PIV_CHECK_HIP(hipMemcpy(d, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
    *dev = (float *)d;
    return PIVLFN_OK;
}

static const pivlfn_tensor *find(const TMap &m, const std::string &name, int d0, int d1, int d2, int d3, int ndim)
{
    auto it = m.find(name);
    if (it == m.end()) {
        set_error("state dict: missing key '%s'", name.c_str());
        return nullptr;
    }
    const pivlfn_tensor *t = it->second;
    const int want[4] = {d0, d1, d2, d3};
    bool ok = t->ndim == ndim && t->data != nullptr;
    for (int i = 0; ok && i < ndim; ++i) ok = t->shape[i] == want[i];
    if (!ok) {
        set_error("state dict: '%s' has the wrong shape (want [%d,%d,%d,%d] ndim %d)", name.c_str(), d0, d1, d2, d3, ndim);
        return nullptr;
    }
    return t;
}

struct SegDef { int creal, cload; int coff = -1; };   // coff: first input channel of this source in the OIHW weight (-1 = running offset)

// OIHW weights -> [chunk][tap][half][cout_pad][4]; chunk = 8 staged input channels of one source.
// Element (chunk, tap, h, n, j) multiplies staged channel 8*chunk_in_seg + 4*h + j of that source; when only one quad
// of the source is left (its 4-channel tail) the chunk is a half chunk: channels 2*h + j, j < 2 (two MFMAs per tap).
static int pack_conv(pivlfn_net *net, const TMap &m, const std::string &name, int cout, int cin, int kh, int kw,
                     const std::vector<SegDef> &segs, ConvW *out)
{
    const pivlfn_tensor *w = find(m, name + ".weight", cout, cin, kh, kw, 4);
    const pivlfn_tensor *b = find(m, name + ".bias", cout, 0, 0, 0, 1);
    if (!w || !b) return PIVLFN_ERR_WEIGHTS;
    int creal = 0, nchunk = 0;
    for (auto &s : segs) { creal += s.creal; nchunk += (s.cload + 7) / 8; }
    if (creal != cin) { set_error("internal: segment channels %d != cin %d for %s", creal, cin, name.c_str()); return PIVLFN_ERR_WEIGHTS; }
    const int taps = kh * kw, cp = rup(cout, 32);
    std::vector<float> pk((size_t)nchunk * taps * 2 * cp * 4, 0.f), bias(cp, 0.f);
    int chunk = 0, run = 0, tail = 0;
    for (size_t si = 0; si < segs.size(); ++si) {
        const SegDef &s = segs[si];
        const int coff = s.coff >= 0 ? s.coff : run;
        if (s.cload % 8 == 4 && si + 1 != segs.size()) {
            set_error("internal: only the last source of %s may end in a 4-channel tail", name.c_str());
            return PIVLFN_ERR_WEIGHTS;
        }
        for (int c0 = 0; c0 < s.cload; c0 += 8, ++chunk) {
            const bool half = s.cload - c0 <= 4;     // 4-channel tail: lane half h holds channels {2h, 2h+1} in slots j = 0, 1
            if (half) tail = 1;
            for (int t = 0; t < taps; ++t)
                for (int h = 0; h < 2; ++h)
                    for (int j = 0; j < (half ? 2 : 4); ++j) {
                        const int c = c0 + (half ? 2 * h : 4 * h) + j;
                        if (c >= s.creal) continue;
                        for (int n = 0; n < cout; ++n)
                            pk[((((size_t)chunk * taps + t) * 2 + h) * cp + n) * 4 + j] =
                                w->data[((size_t)n * cin + coff + c) * taps + t];
                    }
        }
        run += s.creal;
    }
    for (int n = 0; n < cout; ++n) bias[n] = b->data[n];
    out->cout = cout; out->cout_pad = cp; out->KH = kh; out->KW = kw; out->nchunk = nchunk; out->tail = tail; out->cin = cin;
    int rc = upload(net, pk, &out->wpk);
    if (rc) return rc;
    {   // the fp16 packing of the same layer (optional reduced-precision mode)
        std::vector<int> cr, cl, co;
        for (auto &sg : segs) { cr.push_back(sg.creal); cl.push_back(sg.cload); co.push_back(sg.coff); }
        std::vector<unsigned short> ph;
        pack_conv_h(w->data, cout, cin, taps, cr.data(), cl.data(), co.data(), (int)segs.size(), ph, &out->nchunk_h);
        void *d = nullptr;
        PIV_CHECK_HIP(hipMalloc(&d, ph.size() * sizeof(unsigned short)));
        net->allocs.push_back(d);
        PIV_CHECK_HIP(hipMemcpy(d, ph.data(), ph.size() * sizeof(unsigned short), hipMemcpyHostToDevice));
        out->wpk_h = d;
    }
    if (kh == 7 && kw == 1 && cin == 32 && cout <= 64 && segs.size() == 1 && segs[0].cload == 32) {   // conv_dist_R.0 of levels 1 and 2
        std::vector<float> pc((size_t)4 * 7 * 2 * 64 * 4, 0.f);
        for (int blk = 0; blk < 4; ++blk)
            for (int ky = 0; ky < 7; ++ky)
                for (int hh = 0; hh < 2; ++hh)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int j = 0; j < 4; ++j) {
                            // 49 channels: block 3 = channel 48 replicated over the slots (the kernel's fourth wave takes it on the vector unit)
                            const int o = (cout == 49 && blk == 3) ? 48 : 16 * blk + (lane & 15), c = 16 * hh + 4 * (lane >> 4) + j;
                            if (o < cout) pc[((((size_t)blk * 7 + ky) * 2 + hh) * 64 + lane) * 4 + j] = w->data[((size_t)o * 32 + c) * 7 + ky];
                        }
        rc = upload(net, pc, &out->wpk_c);
        if (rc) return rc;
    }
    if (kh == 1 && kw == 7 && cin == 49 && cout == 49 && segs.size() == 1 && segs[0].cload == 52) {   // conv_dist_R.1 of levels 1 and 2
        std::vector<float> pr((size_t)4 * 7 * 3 * 64 * 4, 0.f), p12((size_t)4 * 7 * 64, 0.f);
        for (int blk = 0; blk < 4; ++blk)
            for (int kx = 0; kx < 7; ++kx)
                for (int lane = 0; lane < 64; ++lane) {
                    // block 3 = output channel 48 replicated over the slots (the kernel's vector path)
                    const int o = blk == 3 ? 48 : 16 * blk + (lane & 15), kq = lane >> 4;
                    for (int g = 0; g < 3; ++g)
                        for (int j = 0; j < 4; ++j)
                            pr[((((size_t)blk * 7 + kx) * 3 + g) * 64 + lane) * 4 + j] = w->data[((size_t)o * 49 + 4 * (kq + 4 * g) + j) * 7 + kx];
                    if (kq == 0) p12[((size_t)blk * 7 + kx) * 64 + lane] = w->data[((size_t)o * 49 + 48) * 7 + kx];
                }
        rc = upload(net, pr, &out->wpk_r);
        if (rc) return rc;
        rc = upload(net, p12, &out->wpk_r12);
        if (rc) return rc;
    }
    if (kh == 3 && kw == 3) {      // 3 x 3: the Winograd-domain packing (used by the stride-1 call sites)
        std::vector<int> cr, cl, co;
        for (auto &sg : segs) { cr.push_back(sg.creal); cl.push_back(sg.cload); co.push_back(sg.coff); }
        std::vector<float> pw;
        pack_conv_w(w->data, cout, cin, cr.data(), cl.data(), co.data(), (int)segs.size(), pw, &out->nchunk_w);
        rc = upload(net, pw, &out->wpk_w);
        if (rc) return rc;
        // F(4x4): 2.25 x the F(2x2) planes per layer -- only where something can launch it (round 4 packed and uploaded it for every
        // 3 x 3 layer of every network although pivlfn_forward never reaches that kernel outside the tools build's knob 13)
#ifdef PIVLFN_TOOLS
        pack_conv_w4(w->data, cout, cin, cr.data(), cl.data(), co.data(), (int)segs.size(), pw, &out->nchunk_w4);
        rc = upload(net, pw, &out->wpk_w4);
        if (rc) return rc;
#endif
        if (conv_wino_b3_supports(cp)) {
            std::vector<unsigned short> pb;
            pack_conv_wb(w->data, cout, cin, cr.data(), cl.data(), co.data(), (int)segs.size(), pb, &out->nstep_wb);
            void *d = nullptr;
            PIV_CHECK_HIP(hipMalloc(&d, pb.size() * sizeof(unsigned short)));
            net->allocs.push_back(d);
            PIV_CHECK_HIP(hipMemcpy(d, pb.data(), pb.size() * sizeof(unsigned short), hipMemcpyHostToDevice));
            out->wpk_wb = d;
        }
    }
    if (conv_split_supports(kh, kw, 1, cp, 6)) {   // the split-operand packing of the same layer (fp32 on the fp16 matrix cores)
        std::vector<int> cr, cl, co;
        for (auto &sg : segs) { cr.push_back(sg.creal); cl.push_back(sg.cload); co.push_back(sg.coff); }
        std::vector<unsigned short> px;
        pack_conv_x(w->data, cout, cin, taps, cr.data(), cl.data(), co.data(), (int)segs.size(), px, &out->nchunk_x, &out->scale_x);
        void *d = nullptr;
        PIV_CHECK_HIP(hipMalloc(&d, px.size() * sizeof(unsigned short)));
        net->allocs.push_back(d);
        PIV_CHECK_HIP(hipMemcpy(d, px.data(), px.size() * sizeof(unsigned short), hipMemcpyHostToDevice));
        out->wpk_x = d;
        // a 4-lane tail (the last source's cload = 4 mod 16): the same channels again, taps folded into K, for the 16-row kernel
        const SegDef &ls = segs.back();
        if (kh == 3 && kw == 3 && ls.cload % 16 == 4) {
            int run = 0;
            for (size_t si = 0; si + 1 < segs.size(); ++si) run += segs[si].creal;
            const int first = (ls.coff >= 0 ? ls.coff : run) + ls.cload - 4;          // first weight channel of the tail lanes
            const int real = std::max(0, std::min(4, ls.creal - (ls.cload - 4)));
            std::vector<unsigned short> pt;
            pack_conv_x_tail(w->data, cout, cin, first, real, out->scale_x, pt);
            void *dt = nullptr;
            PIV_CHECK_HIP(hipMalloc(&dt, pt.size() * sizeof(unsigned short)));
            net->allocs.push_back(dt);
            PIV_CHECK_HIP(hipMemcpy(dt, pt.data(), pt.size() * sizeof(unsigned short), hipMemcpyHostToDevice));
            out->wtail_x = dt;
        }
    }
    return upload(net, bias, &out->bias);
}

static int pack_dw(pivlfn_net *net, const TMap &m, const std::string &name, int C, int cpad, float **dev)
{
    const pivlfn_tensor *w = find(m, name, C, 1, 4, 4, 4);
    if (!w) return PIVLFN_ERR_WEIGHTS;
    std::vector<float> h((size_t)cpad * 16, 0.f);      // [16 taps][cpad channels]
    for (int c = 0; c < C; ++c)
        for (int t = 0; t < 16; ++t) h[(size_t)t * cpad + c] = w->data[(size_t)c * 16 + t];
    return upload(net, h, dev);
}

// Flow-head weights for conv_head.hip: OIHW [2,32,k,k] -> [tap][channel quad][4][output] (the two outputs of a channel adjacent:
// one 64-bit scalar operand of a packed fp32 fma)
static int pack_head(pivlfn_net *net, const TMap &m, const std::string &name, int k, float **dev, float bias[2])
{
    const pivlfn_tensor *w = find(m, name + ".weight", 2, 32, k, k, 4);
    const pivlfn_tensor *b = find(m, name + ".bias", 2, 0, 0, 0, 1);
    if (!w || !b) return PIVLFN_ERR_WEIGHTS;
    // [k*k][8][4][2] for the vector kernels, followed by the A fragments of the matrix-core head (conv_head_mfma_kernel):
    // [ky][half h][lane 64][4]: lane = slot n (= 8 o + kx) + 16 kq, element j multiplies channel 16 h + 4 kq + j (zero for kx >= k)
    std::vector<float> h((size_t)k * k * 64 + (size_t)k * 2 * 64 * 4, 0.f);
    for (int t = 0; t < k * k; ++t)
        for (int q = 0; q < 8; ++q)
            for (int o = 0; o < 2; ++o)
                for (int j = 0; j < 4; ++j)
                    h[(((size_t)t * 8 + q) * 4 + j) * 2 + o] = w->data[((size_t)o * 32 + 4 * q + j) * k * k + t];
    for (int ky = 0; ky < k; ++ky)
        for (int hh = 0; hh < 2; ++hh)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 4; ++j) {
                    const int n = lane & 15, kq = lane >> 4, o = n >> 3, kx = n & 7, c = 16 * hh + 4 * kq + j;
                    if (kx < k) h[(size_t)k * k * 64 + (((size_t)ky * 2 + hh) * 64 + lane) * 4 + j] = w->data[((size_t)o * 32 + c) * k * k + ky * k + kx];
                }
    bias[0] = b->data[0];
    bias[1] = b->data[1];
    return upload(net, h, dev);
}

int net_destroy(pivlfn_net *net)
{
    if (!net) return PIVLFN_OK;
    for (void *p : net->allocs) (void)hipFree(p);
    for (hipEvent_t e : net->ev) (void)hipEventDestroy(e);
    for (hipEvent_t e : net->ev_fork) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : net->ev_join) if (e) (void)hipEventDestroy(e);
    if (net->side) (void)hipStreamDestroy(net->side);
    delete net;
    return PIVLFN_OK;
}

// Per timed launch of the chosen level's warp+correlation: events [0,1] = start/stop of the dispatch itself
// (hipExtLaunchKernelGGL), events [2,3] = a plain hipEventRecord pair around it.  Read back after the timed region.
int net_profile_enable(pivlfn_net *net, int level)
{
    PIV_REQUIRE(net && level >= 0 && level <= 6, "profile_enable: bad arguments");
    net->prof_level = level;
    if (level && net->ev.empty()) {
        net->ev.resize(4 * 4096);
        for (auto &e : net->ev) PIV_CHECK_HIP(hipEventCreate(&e));
    }
    net->ev_used = 0;
    net->ev_dropped = 0;
    return PIVLFN_OK;
}

int net_profile_read(pivlfn_net *net, double *ms, double *ms_empty, long *launches, int reset)
{
    PIV_REQUIRE(net && ms && ms_empty && launches, "profile_read: null argument");
    double tot = 0.0, empty = 0.0;
    for (size_t i = 0; i + 3 < net->ev_used; i += 4) {
        PIV_CHECK_HIP(hipEventSynchronize(net->ev[i + 3]));
        float t = 0.f, e = 0.f;
        PIV_CHECK_HIP(hipEventElapsedTime(&t, net->ev[i], net->ev[i + 1]));
        PIV_CHECK_HIP(hipEventElapsedTime(&e, net->ev[i + 2], net->ev[i + 3]));
        tot += t;
        empty += e;
    }
    *ms = tot;
    *ms_empty = empty;
    *launches = (long)(net->ev_used / 4);
    if (reset) { net->ev_used = 0; net->ev_dropped = 0; }
    return PIVLFN_OK;
}

int net_create(const pivlfn_tensor *tensors, int n, float starting_scale, int lowest, const float mean[6], pivlfn_net **out)
{
    PIV_REQUIRE(tensors && n > 0 && out && mean, "create: null argument");
    PIV_REQUIRE(lowest >= 1 && lowest <= 6, "create: lowest_level=%d out of range", lowest);
    TMap m;
    for (int i = 0; i < n; ++i) {
        PIV_REQUIRE(tensors[i].name, "create: tensor %d has no name", i);
        m[tensors[i].name] = &tensors[i];
    }
    pivlfn_net *net = new pivlfn_net();
    net->lowest = lowest;
    for (int L = 0; L < 7; ++L) net->scale[L] = starting_scale / (float)(1 << L);     // src/models.py:61-63
    for (int i = 0; i < 6; ++i) net->mean[i] = mean[i];
    if (m.count("NetE_M.0.conv_M.10.weight")) {          // LiteFlowNet2 layout: five hidden layers per stack
        net->nstack = 5;
        const int w2[5] = {128, 128, 96, 64, 32};
        for (int j = 0; j < 5; ++j) net->width[j] = w2[j];
    }
#define TRY(expr) do { int _rc = (expr); if (_rc) { net_destroy(net); return _rc; } } while (0)
    // NetC (src/models.py:70-106); conv1 reads the 4-lane padded image
    struct { const char *name; int cout, cin, k; } nc[10] = {
        {"NetC.conv1.0", 32, 3, 7}, {"NetC.conv2.0", 32, 32, 3}, {"NetC.conv2.2", 32, 32, 3}, {"NetC.conv2.4", 32, 32, 3},
        {"NetC.conv3.0", 64, 32, 3}, {"NetC.conv3.2", 64, 64, 3}, {"NetC.conv4.0", 96, 64, 3}, {"NetC.conv4.2", 96, 96, 3},
        {"NetC.conv5.0", 128, 96, 3}, {"NetC.conv6.0", 192, 128, 3}};
    for (int i = 0; i < 10; ++i)
        TRY(pack_conv(net, m, nc[i].name, nc[i].cout, nc[i].cin, nc[i].k, nc[i].k, {{nc[i].cin, rup(nc[i].cin, 4)}}, &net->netc[i]));
    // NetC_ext (src/models.py:309-311, 353-355): idx = L-1; NetC_ext[idx-1], python negative index for L1
    const int n_ext = lowest <= 2 ? 2 - (lowest - 1) : 0;
    for (int L = lowest; L <= 2; ++L) {
        int j = (L - 1) - 1;
        if (j < 0) j += n_ext;
        TRY(pack_conv(net, m, "NetC_ext." + std::to_string(j) + ".conv_ext.0", 64, 32, 1, 1, {{32, 32}}, &net->ext[L]));
    }
    for (int L = lowest; L <= 6; ++L) {
        const int i = L - lowest, k = K_LEVEL[L], cm = C_MATCH[L];
        LevelW &lw = net->lv[L];
        const std::string pm = "NetE_M." + std::to_string(i) + ".", ps = "NetE_S." + std::to_string(i) + ".",
                          pr = "NetE_R." + std::to_string(i) + ".";
        if (L != 6) TRY(pack_dw(net, m, pm + "upConv_M.weight", 2, 4, &lw.upconv));
        if (L < 4) TRY(pack_dw(net, m, pm + "upCorr_M.weight", 49, 56, &lw.upcorr));
        {
            int cin = 49;
            for (int j = 0; j < net->nstack; ++j) {
                const int wd = net->width[j];
                const std::string nm = pm + "conv_M." + std::to_string(2 * j);
                if (j == 0) TRY(pack_conv(net, m, nm, wd, 49, 3, 3, {{49, 52}}, &lw.M[0]));
                else TRY(pack_conv(net, m, nm, wd, cin, 3, 3, {{cin, cin}}, &lw.M[j]));
                cin = wd;
            }
            const std::string hm = pm + "conv_M." + std::to_string(2 * net->nstack);
            TRY(pack_conv(net, m, hm, 2, 32, k, k, {{32, 32}}, &lw.M[net->nstack]));
            TRY(pack_head(net, m, hm, k, &lw.headM, lw.hbM));
            cin = 2 * cm + 2;
            for (int j = 0; j < net->nstack; ++j) {
                const int wd = net->width[j];
                const std::string nm = ps + "conv_S." + std::to_string(2 * j);
                if (j == 0) TRY(pack_conv(net, m, nm, wd, 2 * cm + 2, 3, 3, {{cm, cm}, {cm, cm}, {2, 4}}, &lw.S[0]));
                else TRY(pack_conv(net, m, nm, wd, cin, 3, 3, {{cin, cin}}, &lw.S[j]));
                cin = wd;
            }
            const std::string hs = ps + "conv_S." + std::to_string(2 * net->nstack);
            TRY(pack_conv(net, m, hs, 2, 32, k, k, {{32, 32}}, &lw.S[net->nstack]));
            TRY(pack_head(net, m, hs, k, &lw.headS, lw.hbS));
        }
        const int cfr = L < 5 ? 128 : C_FEAT[L];
        if (L < 5) TRY(pack_conv(net, m, pr + "moduleFeat.0", 128, C_FEAT[L], 1, 1, {{C_FEAT[L], C_FEAT[L]}}, &lw.feat));
        TRY(pack_conv(net, m, pr + "conv_R.0", 128, 3 + cfr, 3, 3, {{cfr, cfr, 3}, {3, 4, 0}}, &lw.R[0]));   // reference order is [norm, rm, feat] (:280)
        TRY(pack_conv(net, m, pr + "conv_R.2", 128, 128, 3, 3, {{128, 128}}, &lw.R[1]));
        TRY(pack_conv(net, m, pr + "conv_R.4", 64, 128, 3, 3, {{128, 128}}, &lw.R[2]));
        TRY(pack_conv(net, m, pr + "conv_R.6", 64, 64, 3, 3, {{64, 64}}, &lw.R[3]));
        TRY(pack_conv(net, m, pr + "conv_R.8", 32, 64, 3, 3, {{64, 64}}, &lw.R[4]));
        TRY(pack_conv(net, m, pr + "conv_R.10", 32, 32, 3, 3, {{32, 32}}, &lw.R[5]));
        const int kk = k * k;
        if (L < 5) {
            TRY(pack_conv(net, m, pr + "conv_dist_R.0", kk, 32, k, 1, {{32, 32}}, &lw.dist0));
            TRY(pack_conv(net, m, pr + "conv_dist_R.1", kk, kk, 1, k, {{kk, rup(kk, 4)}}, &lw.dist1));
        } else {
            TRY(pack_conv(net, m, pr + "conv_dist_R.0", kk, 32, k, k, {{32, 32}}, &lw.dist0));
        }
        const pivlfn_tensor *wx = find(m, pr + "moduleScaleX.weight", 1, kk, 1, 1, 4), *bx = find(m, pr + "moduleScaleX.bias", 1, 0, 0, 0, 1);
        const pivlfn_tensor *wy = find(m, pr + "moduleScaleY.weight", 1, kk, 1, 1, 4), *by = find(m, pr + "moduleScaleY.bias", 1, 0, 0, 0, 1);
        if (!wx || !bx || !wy || !by) { net_destroy(net); return PIVLFN_ERR_WEIGHTS; }
        TRY(upload(net, std::vector<float>(wx->data, wx->data + kk), &lw.wx));
        TRY(upload(net, std::vector<float>(wy->data, wy->data + kk), &lw.wy));
        lw.bx = bx->data[0];
        lw.by = by->data[0];
    }
    if (lowest == 1) {      // the two 1 x 1 layers that read NetC.conv1's output at level 1, in the fragment order of Conv1Fuse
        int j = -1;
        if (j < 0) j += n_ext;                       // NetC_ext index of level 1 (python negative index, as above)
        const pivlfn_tensor *we = find(m, "NetC_ext." + std::to_string(j) + ".conv_ext.0.weight", 64, 32, 1, 1, 4);
        const pivlfn_tensor *be = find(m, "NetC_ext." + std::to_string(j) + ".conv_ext.0.bias", 64, 0, 0, 0, 1);
        const pivlfn_tensor *wf = find(m, "NetE_R." + std::to_string(1 - lowest) + ".moduleFeat.0.weight", 128, 32, 1, 1, 4);
        const pivlfn_tensor *bfe = find(m, "NetE_R." + std::to_string(1 - lowest) + ".moduleFeat.0.bias", 128, 0, 0, 0, 1);
        if (!we || !be || !wf || !bfe) { net_destroy(net); return PIVLFN_ERR_WEIGHTS; }
        std::vector<float> w11((size_t)6 * 4 * 64 * 4), b11(192);
        for (int blk = 0; blk < 6; ++blk)
            for (int g = 0; g < 4; ++g)
                for (int lane = 0; lane < 64; ++lane)
                    for (int e = 0; e < 4; ++e) {
                        const int c = 8 * g + 4 * (lane >> 5) + e, o = 32 * (blk < 2 ? blk : blk - 2) + (lane & 31);
                        w11[(((size_t)blk * 4 + g) * 64 + lane) * 4 + e] = blk < 2 ? we->data[(size_t)o * 32 + c] : wf->data[(size_t)o * 32 + c];
                    }
        for (int o = 0; o < 64; ++o) b11[o] = be->data[o];
        for (int o = 0; o < 128; ++o) b11[64 + o] = bfe->data[o];
        TRY(upload(net, w11, &net->fuse1_w));
        TRY(upload(net, b11, &net->fuse1_b));
    }
#undef TRY
    if (hipStreamCreateWithFlags(&net->side, hipStreamNonBlocking) != hipSuccess) {
        set_error("create: side stream creation failed");
        net_destroy(net);
        return PIVLFN_ERR_HIP;
    }
    for (int L = lowest; L <= 6; ++L)
        if (hipEventCreateWithFlags(&net->ev_join[L], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&net->ev_fork[L], hipEventDisableTiming) != hipSuccess) {
            set_error("create: event creation failed");
            net_destroy(net);
            return PIVLFN_ERR_HIP;
        }
    *out = net;
    return PIVLFN_OK;
}

// ---- stand-alone convolution layer (tests, micro-benchmarks) ---------------------------------------------------------
int conv_create(const float *weight, const float *bias, int cout, int cin, int kh, int kw, pivlfn_conv **out)
{
    PIV_REQUIRE(weight && bias && out && cout > 0 && cin > 0 && kh > 0 && kw > 0, "conv_create: bad arguments");
    pivlfn_tensor t[2];
    t[0].name = "c.weight"; t[0].data = weight; t[0].ndim = 4;
    t[0].shape[0] = cout; t[0].shape[1] = cin; t[0].shape[2] = kh; t[0].shape[3] = kw;
    t[1].name = "c.bias"; t[1].data = bias; t[1].ndim = 1;
    t[1].shape[0] = cout; t[1].shape[1] = t[1].shape[2] = t[1].shape[3] = 0;
    TMap m;
    m["c.weight"] = &t[0];
    m["c.bias"] = &t[1];
    pivlfn_conv *c = new pivlfn_conv();
    c->owner = new pivlfn_net();
    c->cin = cin;
    int rc = pack_conv(c->owner, m, "c", cout, cin, kh, kw, {{cin, rup(cin, 4)}}, &c->cw);
    if (!rc && cout == 2 && cin == 32 && kh == kw && (kh == 3 || kh == 5 || kh == 7)) rc = pack_head(c->owner, m, "c", kh, &c->head, c->hb);
    if (!rc) {
        void *d = nullptr;
        if (hipMalloc(&d, KSPLIT_FLOATS * sizeof(float)) != hipSuccess) { set_error("conv_create: scratch allocation failed"); rc = PIVLFN_ERR_HIP; }
        else { c->owner->allocs.push_back(d); c->scratch = (float *)d; }
    }
    if (rc) { net_destroy(c->owner); delete c; return rc; }
    *out = c;
    return PIVLFN_OK;
}

int conv_destroy(pivlfn_conv *c)
{
    if (!c) return PIVLFN_OK;
    net_destroy(c->owner);
    delete c;
    return PIVLFN_OK;
}

int conv_forward(const pivlfn_conv *c, const float *x, int x_stride, float *y, int y_stride, const float *res, int res_stride,
                 int B, int H, int W, int stride, int pad_y, int pad_x, int leaky, hipStream_t st)
{
    PIV_REQUIRE(c && x && y, "conv2d: null argument");
    PIV_REQUIRE(x_stride % 4 == 0 && x_stride >= rup(c->cin, 4), "conv2d: x_stride=%d must be a multiple of 4 and >= %d", x_stride, rup(c->cin, 4));
    PIV_REQUIRE(y_stride >= c->cw.cout, "conv2d: y_stride=%d < cout=%d", y_stride, c->cw.cout);
    PIV_REQUIRE(stride >= 1 && pad_y >= 0 && pad_x >= 0 && H + 2 * pad_y >= c->cw.KH && W + 2 * pad_x >= c->cw.KW, "conv2d: bad geometry");
    ConvParams p;
    memset(&p, 0, sizeof(p));
    p.seg[0] = ConvSeg{x, rup(c->cin, 4), x_stride};
    p.nseg = 1;
    p.wpk = c->cw.wpk; p.bias = c->cw.bias; p.out = y; p.out_stride = y_stride;
    p.cout_store = std::min(rup(c->cw.cout, 4), y_stride);
    p.cout_pad = c->cw.cout_pad; p.res = res; p.res_stride = res_stride;
    p.B = B; p.H = H; p.W = W; p.KH = c->cw.KH; p.KW = c->cw.KW; p.S = stride; p.padY = pad_y; p.padX = pad_x;
    p.Ho = (H + 2 * pad_y - c->cw.KH) / stride + 1;
    p.Wo = (W + 2 * pad_x - c->cw.KW) / stride + 1;
    p.nchunk = c->cw.nchunk; p.tail = c->cw.tail; p.lrelu = leaky; p.cin_real = c->cw.cin;
    p.scratch = c->scratch; p.scratch_floats = KSPLIT_FLOATS;
    // the (7 x 1) distance convolution on >= 256 x 256 images: the kernel pivlfn_forward uses for it in the fp32 mode
    if (c->cw.wpk_c && !res && !leaky && stride == 1 && pad_y == 3 && pad_x == 0 && (long)H * W >= 256 * 256 && p.cout_store % 4 == 0 &&
        (long)H * W * std::max(x_stride, y_stride) * 4 < (1L << 31) && !(PIV_KNOB(1) & 65536))
        return launch_conv_col7(x, x_stride, c->cw.wpk_c, c->cw.bias, y, y_stride, p.cout_store, c->cw.cout == 49, B, H, W, st);
    if (c->cw.wpk_r && !res && !leaky && stride == 1 && pad_y == 0 && pad_x == 3 && (long)H * W >= 256 * 256 && x_stride >= 52 && y_stride >= 52 &&
        (long)H * W * std::max(x_stride, y_stride) * 4 < (1L << 31) && !(PIV_KNOB(1) & 65536))
        return launch_conv_row7(x, x_stride, c->cw.wpk_r, c->cw.wpk_r12, c->cw.bias, y, y_stride, B, H, W, st);
    return launch_conv(p, st);
}

// One Conv2d over the channel concatenation of up to three sources (torch.cat + Conv2d, src/models.py:165-187, 209-217, 280: the
// front layers of Matching / Subpixel / Regularization), through the same dispatch as pivlfn_forward's fp32 mode: multi-source
// staging of the direct and the Winograd kernel for per-layer checks.
int conv_create_cat(const float *weight, const float *bias, int cout, int nsrc, const int *channels, int kh, int kw, pivlfn_conv **out)
{
    PIV_REQUIRE(weight && bias && out && channels && cout > 0 && nsrc >= 1 && nsrc <= 3 && kh > 0 && kw > 0, "conv_create_cat: bad arguments");
    int cin = 0;
    std::vector<SegDef> segs;
    for (int i = 0; i < nsrc; ++i) {
        PIV_REQUIRE(channels[i] > 0, "conv_create_cat: source %d has %d channels", i, channels[i]);
        segs.push_back(SegDef{channels[i], rup(channels[i], 4)});
        cin += channels[i];
    }
    pivlfn_tensor t[2];
    t[0].name = "c.weight"; t[0].data = weight; t[0].ndim = 4;
    t[0].shape[0] = cout; t[0].shape[1] = cin; t[0].shape[2] = kh; t[0].shape[3] = kw;
    t[1].name = "c.bias"; t[1].data = bias; t[1].ndim = 1;
    t[1].shape[0] = cout; t[1].shape[1] = t[1].shape[2] = t[1].shape[3] = 0;
    TMap m;
    m["c.weight"] = &t[0];
    m["c.bias"] = &t[1];
    pivlfn_conv *c = new pivlfn_conv();
    c->owner = new pivlfn_net();
    c->cin = cin;
    c->nsrc = nsrc;
    for (int i = 0; i < nsrc; ++i) c->src_real[i] = channels[i];
    const int rc = pack_conv(c->owner, m, "c", cout, cin, kh, kw, segs, &c->cw);
    if (rc) { net_destroy(c->owner); delete c; return rc; }
    *out = c;
    return PIVLFN_OK;
}

int conv_head_forward(const pivlfn_conv *c, const float *x, const float *res4, float *out4, int B, int H, int W, hipStream_t st)
{
    PIV_REQUIRE(c && c->head, "conv_head: the layer is not a 32->2 kxk flow head");
    return launch_conv_head(x, c->head, c->hb[0], c->hb[1], res4, out4, B, H, W, c->cw.KH, st);
}

// ---- workspace plan -------------------------------------------------------------------------------------------
struct Plan {
    size_t off = 0;
    char *base = nullptr;
    float *take(size_t floats)
    {
        float *p = base ? reinterpret_cast<float *>(base + off) : nullptr;
        off += (floats * sizeof(float) + 255) / 256 * 256;
        return p;
    }
};

struct Buffers {
    float *img[7], *feat[7], *ext[3], *sa, *sb;
    float *flowA, *flowB, *flow_up, *flowM, *flowS, *corr, *corr_up, *t128a, *t128b, *t64a, *t64b, *t32a, *t32b,
        *f2w, *featR[7], *misc4, *d1, *dist, *partial, *mean, *ksplit;
};

static void plan(const pivlfn_net *net, int B, int H, int W, Plan &pl, Buffers &bf)
{
    int h[7], w[7];
    for (int L = 1; L <= 6; ++L) { h[L] = H >> (L - 1); w[L] = W >> (L - 1); }
    const size_t N2 = 2 * (size_t)B;
    for (int L = 1; L <= 6; ++L) bf.img[L] = pl.take(N2 * h[L] * w[L] * 4);
    for (int L = 1; L <= 6; ++L) bf.feat[L] = pl.take(N2 * h[L] * w[L] * C_FEAT[L]);
    for (int L = 1; L <= 2; ++L) bf.ext[L] = L >= net->lowest ? pl.take(N2 * h[L] * w[L] * 64) : nullptr;
    bf.sa = pl.take(N2 * h[2] * w[2] * 32);
    bf.sb = pl.take(N2 * h[2] * w[2] * 32);
    const int ll = net->lowest;
    const size_t px = (size_t)B * h[ll] * w[ll];
    size_t f2w = 0;
    for (int L = ll; L <= 6; ++L) f2w = std::max(f2w, (size_t)B * h[L] * w[L] * C_MATCH[L]);
    bf.flowA = pl.take(px * 4); bf.flowB = pl.take(px * 4); bf.flow_up = pl.take(px * 4);
    bf.flowM = pl.take(px * 4); bf.flowS = pl.take(px * 4);
    bf.corr = pl.take(px * 56); bf.corr_up = pl.take(px * 56);
    bf.t128a = pl.take(px * 128); bf.t128b = pl.take(px * 128);
    bf.t64a = pl.take(px * 64); bf.t64b = pl.take(px * 64);
    bf.t32a = pl.take(px * 32); bf.t32b = pl.take(px * 32);
    bf.f2w = pl.take(f2w);
    for (int L = 1; L <= 6; ++L) bf.featR[L] = (L >= ll && L < 5) ? pl.take((size_t)B * h[L] * w[L] * 128) : nullptr;   // one per level: filled on the side stream
    bf.misc4 = pl.take(px * 4);
    bf.d1 = pl.take(px * 56); bf.dist = pl.take(px * 56);
    bf.ksplit = pl.take(KSPLIT_FLOATS * 2 * B);     // per image (NetC runs 2B images): the split never depends on the batch
    bf.partial = pl.take((size_t)B * flow_mean_partials(0) * 2);
    bf.mean = pl.take((size_t)B * 2);
}

size_t net_workspace_bytes(const pivlfn_net *net, int B, int H, int W)
{
    Plan pl; Buffers bf;
    plan(net, B, H, W, pl, bf);
    return pl.off;
}

size_t net_levels_floats(const pivlfn_net *net, int B, int H, int W)
{
    size_t n = 0;
    for (int L = net->lowest; L <= 6; ++L) n += (size_t)3 * B * 2 * (H >> (L - 1)) * (W >> (L - 1));
    return n;
}

static thread_local int t_precision = 0;      // set by net_forward for the duration of one forward
static thread_local bool t_no_b3 = false;      // PIVLFN_PRECISION_F32_WINO_MFMA32: t_precision 0 with every Winograd layer on the fp32 instruction
static thread_local float *t_scratch = nullptr;          // split-K scratch of the forward in progress (main stream only)
static thread_local hipStream_t t_side = nullptr;

// in16: bit i set = source i holds fp16 elements; out16: the output is stored as fp16.  Both are only ever non-zero for layers
// that run on the fp16 kernel (net_forward's `h16` uses the same size test as below).
static int conv(const ConvW &cw, std::initializer_list<ConvSeg> segs, float *out, int out_stride, int cout_store,
                const float *res, int res_stride, int lrelu, int B, int H, int W, int S, int padY, int padX, hipStream_t st,
                int in16 = 0, int out16 = 0)
{
    const int Ho = (H + 2 * padY - cw.KH) / S + 1, Wo = (W + 2 * padX - cw.KW) / S + 1;
    // fp16 mode: every residual-free conv whose output grid is at least 64x64 (smaller levels are launch-latency-bound and
    // stay on the fp32 kernel); activations stay fp32 in HBM, operands are rounded to fp16 while they are staged.
    if (t_precision == 1 && !res && (long)Ho * Wo >= 64 * 64) {
        ConvParamsH q;
        memset(&q, 0, sizeof(q));
        int i = 0;
        for (auto &sg : segs) { q.seg[i] = ConvSegH{sg.ptr, sg.cload, sg.stride, (in16 >> i) & 1}; ++i; }
        q.nseg = i;
        q.wpk = cw.wpk_h; q.bias = cw.bias; q.out = out; q.out_stride = out_stride; q.cout_store = cout_store;
        q.cout_pad = cw.cout_pad; q.out_f16 = out16;
        q.B = B; q.H = H; q.W = W; q.Ho = Ho; q.Wo = Wo;
        q.KH = cw.KH; q.KW = cw.KW; q.S = S; q.padY = padY; q.padX = padX;
        q.nchunk = cw.nchunk_h; q.lrelu = lrelu;
        return launch_conv_h(q, st);
    }
    PIV_REQUIRE(!in16 && !out16, "internal: fp16 activations routed to the fp32 conv kernel");
    // split modes: every residual-free conv the split kernel covers, with an output grid of at least 64x64 per image (the three-term
    // kernel has 4-row tiles and split-K for the small grids; below 64x64 the layers are a dependent chain of ~12 us launches on
    // either kernel).  The six-term kernel has neither and keeps the 256x256 bound.  Per image: the choice never depends on the
    // batch (tools/split_threshold.py: 1024^2, 512^2 and 256^2 inputs).
    if ((t_precision == 2 || t_precision == 3) && !res && cw.wpk_x && conv_split_supports(cw.KH, cw.KW, S, cw.cout_pad, t_precision == 3 ? 3 : 6) &&
        (long)Ho * Wo >= (PIV_KNOB(11) ? PIV_KNOB(11) : (t_precision == 3 ? 64 * 64 : 256 * 256))) {
        ConvParamsX q;
        memset(&q, 0, sizeof(q));
        int i = 0;
        for (auto &sg : segs) q.seg[i++] = sg;
        q.nseg = i;
        q.wpk = cw.wpk_x; q.wtail = cw.wtail_x; q.bias = cw.bias; q.out = out; q.out_stride = out_stride; q.cout_store = cout_store;
        q.cout_pad = cw.cout_pad; q.out_scale = cw.scale_x; q.terms = t_precision == 3 ? 3 : 6;
        q.B = B; q.H = H; q.W = W; q.Ho = Ho; q.Wo = Wo;
        q.KH = cw.KH; q.KW = cw.KW; q.S = S; q.padY = padY; q.padX = padX;
        q.nchunk = cw.nchunk_x; q.lrelu = lrelu;
        q.scratch = (t_side && st == t_side) ? nullptr : t_scratch;      // one scratch area: the side stream never splits
        q.scratch_floats = KSPLIT_FLOATS * B;
        return launch_conv_x(q, st);
    }
    // fp32 mode: the (7 x 1) distance convolution of levels 1 and 2 on its streaming matrix-core kernel (per image: >= 256 x 256)
    if (t_precision == 0 && !res && cw.wpk_c && cw.KH == 7 && cw.KW == 1 && S == 1 && padY == 3 && padX == 0 && !lrelu &&
        segs.size() == 1 && segs.begin()->cload == 32 && (long)Ho * Wo >= 256 * 256 && cout_store % 4 == 0 &&
        (long)H * W * std::max(segs.begin()->stride, out_stride) * 4 < (1L << 31) && cw.cout_pad <= 64)
        return launch_conv_col7(segs.begin()->ptr, segs.begin()->stride, cw.wpk_c, cw.bias, out, out_stride, cout_store, cw.cout == 49, B, H, W, st);
    if (t_precision == 0 && !res && cw.wpk_r && cw.KH == 1 && cw.KW == 7 && S == 1 && padY == 0 && padX == 3 && !lrelu &&
        segs.size() == 1 && segs.begin()->cload == 52 && segs.begin()->stride >= 52 && out_stride >= 52 && cout_store == 52 &&
        (long)Ho * Wo >= 256 * 256 && (long)H * W * std::max(segs.begin()->stride, out_stride) * 4 < (1L << 31) && !(PIV_KNOB(1) & 65536))
        return launch_conv_row7(segs.begin()->ptr, segs.begin()->stride, cw.wpk_r, cw.wpk_r12, cw.bias, out, out_stride, B, H, W, st);
    // fp32 mode: the 3 x 3 / stride 1 layers by Winograd F(2x2, 3x3) on the fp32 matrix instruction (conv_wino.hip) from a
    // 64 x 64 grid per image up (a 32 x 32 grid is 32 workgroups with the whole K loop each: the split-K direct kernel is faster);
    // the bound is per image, never a function of the batch
    if (t_precision == 0 && !res && cw.wpk_w && conv_wino_supports(cw.KH, cw.KW, S, padY, padX) &&
        (long)Ho * Wo >= (PIV_KNOB(12) ? PIV_KNOB(12) : 64 * 64) && cout_store % 4 == 0) {
        ConvParamsW q;
        memset(&q, 0, sizeof(q));
        int i = 0, cl = 0;
        for (auto &sg : segs) { q.seg[i++] = sg; cl += sg.cload; }
        q.nseg = i;
        // Default fp32 mode: layers with whole 64-channel output groups and at least 48 staged input channels run the same Winograd
        // algorithm with every operand split exactly into three bf16 pieces on the 16-bit matrix cores (conv_wino_b3.hip; all 24
        // significand bits, error against float64 at or below the fp32 instruction's: tests/test_gpu_wino_b3.py) -- 1.08-1.3 x the
        // speed of the fp32-instruction kernel on those layers at 256^2 ... 1024^2 (conv_M.0's 49 channels, four K steps: 1.08-1.16);
        // 32-channel inputs (two steps per tile: the tile's fixed cost decides, 1.0 x) and the 32- and 96-channel outputs stay on
        // conv_wino.hip.  Per layer shape, never per batch.  PIVLFN_PRECISION_F32_WINO_MFMA32 keeps
        // every layer on the fp32 instruction.
        // From 256 x 256 outputs per image: its persistent workgroups (one per CU, 16 x 16 pixels x 64 channels per tile) need at least a
        // tile per CU; the 128 x 128 layers of level 4 took 23-64 us on it against 12-25 us on conv_wino.hip.
        if (!t_no_b3 && cw.wpk_wb && conv_wino_b3_supports(cw.cout_pad) && cl >= 48 && (long)Ho * Wo >= 256 * 256 && !(PIV_KNOB(1) & 2097152) &&
            (long)16 * W * out_stride * 4 < (1L << 31)) {
            q.wpk_b = cw.wpk_wb; q.bias = cw.bias; q.out = out; q.out_stride = out_stride; q.cout_store = cout_store;
            q.cout_pad = cw.cout_pad;
            q.B = B; q.H = H; q.W = W; q.nchunk = cw.nstep_wb; q.lrelu = lrelu; q.terms = 6;
            return launch_conv_wb(q, st);
        }
        q.wpk = cw.wpk_w; q.bias = cw.bias; q.out = out; q.out_stride = out_stride; q.cout_store = cout_store;
        q.cout_pad = cw.cout_pad;
        q.B = B; q.H = H; q.W = W; q.nchunk = cw.nchunk_w; q.lrelu = lrelu;
        // F(4x4, 3x3) is not used by pivlfn_forward: 1.78x fewer matrix instructions, but its 6x6 transforms, 106 KB of LDS (one
        // workgroup per CU) and lockstep of 12 waves leave it at 0.98x of F(2x2) on 128->128 and 0.68x on 32->32 at 1024 x 1024
        // (DESIGN.md 4.2c).  The tools build can switch it in from knob 13 output pixels per image up, for A/B runs of the forward.
#ifdef PIVLFN_TOOLS
        if (cw.wpk_w4 && PIV_KNOB(13) > 0 && (long)Ho * Wo >= PIV_KNOB(13)) {
            q.wpk = cw.wpk_w4; q.nchunk = cw.nchunk_w4;
            return launch_conv_w4(q, st);
        }
#endif
        return launch_conv_w(q, st);
    }
    ConvParams p;
    memset(&p, 0, sizeof(p));
    int i = 0;
    for (auto &s : segs) p.seg[i++] = s;
    p.nseg = i;
    p.wpk = cw.wpk; p.bias = cw.bias; p.out = out; p.out_stride = out_stride; p.cout_store = cout_store;
    p.cout_pad = cw.cout_pad; p.res = res; p.res_stride = res_stride;
    p.B = B; p.H = H; p.W = W;
    p.KH = cw.KH; p.KW = cw.KW; p.S = S; p.padY = padY; p.padX = padX;
    p.Ho = Ho;
    p.Wo = Wo;
    p.nchunk = cw.nchunk; p.tail = cw.tail; p.lrelu = lrelu; p.cin_real = cw.cin;
    p.scratch = (t_side && st == t_side) ? nullptr : t_scratch;      // one scratch area: the side stream never splits
    p.scratch_floats = KSPLIT_FLOATS * B;
    return launch_conv(p, st);
}

int net_set_precision(pivlfn_net *net, int precision)
{
    PIV_REQUIRE(net && precision >= 0 && precision <= 5, "set_precision: 0 (fp32: Winograd with exactly split operands / fp32 instruction), 1 (fp16 multiplicands), 2 (fp32 by exact fp16 splitting), 3 (three-term splitting), 4 (fp32 instruction, direct convolution only) or 5 (fp32 instruction, Winograd) expected");
    net->precision = precision;
    return PIVLFN_OK;
}

// Standalone layer in the fp16 mode (tests, tools): x / y element types chosen per call.
int conv_forward_h(const pivlfn_conv *c, const void *x, int x_stride, int x_f16, void *y, int y_stride, int y_f16,
                   int B, int H, int W, int stride, int pad_y, int pad_x, int leaky, hipStream_t st)
{
    PIV_REQUIRE(c && x && y, "conv2d_f16: null argument");
    const int g = x_f16 ? 8 : 4;
    PIV_REQUIRE(x_stride % g == 0 && x_stride >= rup(c->cin, g), "conv2d_f16: x_stride=%d must be a multiple of %d and >= %d", x_stride, g, rup(c->cin, g));
    PIV_REQUIRE(y_stride % 4 == 0 && y_stride >= c->cw.cout, "conv2d_f16: y_stride=%d must be a multiple of 4 and >= cout=%d", y_stride, c->cw.cout);
    PIV_REQUIRE(stride >= 1 && pad_y >= 0 && pad_x >= 0 && H + 2 * pad_y >= c->cw.KH && W + 2 * pad_x >= c->cw.KW, "conv2d_f16: bad geometry");
    ConvParamsH q;
    memset(&q, 0, sizeof(q));
    q.seg[0] = ConvSegH{x, rup(c->cin, g), x_stride, x_f16};
    q.nseg = 1;
    q.wpk = c->cw.wpk_h; q.bias = c->cw.bias; q.out = y; q.out_stride = y_stride;
    q.cout_store = std::min(rup(c->cw.cout, 4), y_stride);
    q.cout_pad = c->cw.cout_pad; q.out_f16 = y_f16;
    q.B = B; q.H = H; q.W = W; q.KH = c->cw.KH; q.KW = c->cw.KW; q.S = stride; q.padY = pad_y; q.padX = pad_x;
    q.Ho = (H + 2 * pad_y - c->cw.KH) / stride + 1;
    q.Wo = (W + 2 * pad_x - c->cw.KW) / stride + 1;
    q.nchunk = c->cw.nchunk_h; q.lrelu = leaky;
    return launch_conv_h(q, st);
}

// Standalone layer on the split-operand kernel (tests, tools): fp32 in, fp32 out.
int conv_forward_x(const pivlfn_conv *c, const float *x, int x_stride, float *y, int y_stride,
                   int B, int H, int W, int stride, int pad_y, int pad_x, int leaky, int terms, hipStream_t st)
{
    PIV_REQUIRE(c && x && y, "conv2d_split: null argument");
    PIV_REQUIRE(terms == 6 || terms == 3, "conv2d_split: terms=%d (6 or 3 partial products per product)", terms);
    PIV_REQUIRE(c->cw.wpk_x && conv_split_supports(c->cw.KH, c->cw.KW, stride, c->cw.cout_pad, terms),
                "conv2d_split: this layer's geometry (k=%dx%d, stride %d, %d-term products) is not covered by the split kernel", c->cw.KH, c->cw.KW, stride, terms);
    PIV_REQUIRE(x_stride % 4 == 0 && x_stride >= rup(c->cin, 4), "conv2d_split: x_stride=%d must be a multiple of 4 and >= %d", x_stride, rup(c->cin, 4));
    PIV_REQUIRE(y_stride % 4 == 0 && y_stride >= c->cw.cout, "conv2d_split: y_stride=%d must be a multiple of 4 and >= cout=%d", y_stride, c->cw.cout);
    PIV_REQUIRE(pad_y >= 0 && pad_x >= 0 && H + 2 * pad_y >= c->cw.KH && W + 2 * pad_x >= c->cw.KW, "conv2d_split: bad geometry");
    ConvParamsX q;
    memset(&q, 0, sizeof(q));
    q.seg[0] = ConvSeg{x, rup(c->cin, 4), x_stride};
    q.nseg = 1;
    q.wpk = c->cw.wpk_x; q.wtail = c->cw.wtail_x; q.bias = c->cw.bias; q.out = y; q.out_stride = y_stride;
    q.cout_store = std::min(rup(c->cw.cout, 4), y_stride);
    q.cout_pad = c->cw.cout_pad; q.out_scale = c->cw.scale_x; q.terms = terms;
    q.B = B; q.H = H; q.W = W; q.KH = c->cw.KH; q.KW = c->cw.KW; q.S = stride; q.padY = pad_y; q.padX = pad_x;
    q.Ho = (H + 2 * pad_y - c->cw.KH) / stride + 1;
    q.Wo = (W + 2 * pad_x - c->cw.KW) / stride + 1;
    q.nchunk = c->cw.nchunk_x; q.lrelu = leaky;
    // one image's worth of split-K scratch: larger batches run image by image, so the split factor -- hence the summation order and
    // the bits of a sample -- is the same whatever the batch (the invariant launch_conv_x states for the network's own calls)
    q.scratch = c->scratch; q.scratch_floats = KSPLIT_FLOATS;
    if (B > 1 && (long)cdiv(q.Wo, 32) * cdiv(q.Ho, 4) * (q.cout_pad / 32) <= 256) {      // the grids launch_conv_x may split
        for (int b = 0; b < B; ++b) {
            ConvParamsX qb = q;
            qb.B = 1;
            qb.seg[0].ptr = x + (size_t)b * H * W * x_stride;
            qb.out = y + (size_t)b * q.Ho * q.Wo * y_stride;
            if (int rc = launch_conv_x(qb, st)) return rc;
        }
        return PIVLFN_OK;
    }
    return launch_conv_x(q, st);
}

// Standalone 3 x 3 / stride 1 / pad 1 layer on the Winograd kernels (tests, tools): fp32 in, fp32 out.  tile = 2: F(2x2, 3x3), 4: F(4x4, 3x3).
int conv_forward_cat(const pivlfn_conv *c, int nsrc, const float *const *x, const int *x_stride, float *y, int y_stride,
                     int B, int H, int W, int leaky, hipStream_t st)
{
    PIV_REQUIRE(c && x && x_stride && y && nsrc == c->nsrc, "conv2d_cat: the layer was created for %d sources", c ? c->nsrc : 0);
    PIV_REQUIRE(B > 0 && H > 0 && W > 0 && c->cw.KH % 2 == 1 && c->cw.KW % 2 == 1, "conv2d_cat: bad shape");
    PIV_REQUIRE(y_stride % 4 == 0 && y_stride >= c->cw.cout, "conv2d_cat: y_stride=%d must be a multiple of 4 and >= cout=%d", y_stride, c->cw.cout);
    ConvSeg sg[3];
    for (int i = 0; i < nsrc; ++i) {
        PIV_REQUIRE(x[i] && x_stride[i] % 4 == 0 && x_stride[i] >= rup(c->src_real[i], 4), "conv2d_cat: source %d: stride %d for %d channels", i, x_stride[i], c->src_real[i]);
        sg[i] = ConvSeg{x[i], rup(c->src_real[i], 4), x_stride[i]};
    }
    t_precision = 0;
    t_no_b3 = false;
    t_side = nullptr;
    t_scratch = nullptr;          // never split: the handle has no per-batch scratch
    const int cs = std::min(rup(c->cw.cout, 4), y_stride);
    switch (nsrc) {
        case 1: return conv(c->cw, {sg[0]}, y, y_stride, cs, nullptr, 0, leaky, B, H, W, 1, c->cw.KH / 2, c->cw.KW / 2, st, 0, 0);
        case 2: return conv(c->cw, {sg[0], sg[1]}, y, y_stride, cs, nullptr, 0, leaky, B, H, W, 1, c->cw.KH / 2, c->cw.KW / 2, st, 0, 0);
        default: return conv(c->cw, {sg[0], sg[1], sg[2]}, y, y_stride, cs, nullptr, 0, leaky, B, H, W, 1, c->cw.KH / 2, c->cw.KW / 2, st, 0, 0);
    }
}

int conv_forward_w(const pivlfn_conv *c, const float *x, int x_stride, float *y, int y_stride, int B, int H, int W, int leaky,
                   hipStream_t st, int tile)
{
    PIV_REQUIRE(tile == 2 || tile == 4, "conv2d_wino: tile=%d (2 or 4)", tile);
    PIV_REQUIRE(c && x && y, "conv2d_wino: null argument");
    PIV_REQUIRE(c->cw.wpk_w, "conv2d_wino: the layer is not 3 x 3 (k=%dx%d)", c->cw.KH, c->cw.KW);
    PIV_REQUIRE(x_stride % 4 == 0 && x_stride >= rup(c->cin, 4), "conv2d_wino: x_stride=%d must be a multiple of 4 and >= %d", x_stride, rup(c->cin, 4));
    PIV_REQUIRE(y_stride % 4 == 0 && y_stride >= c->cw.cout, "conv2d_wino: y_stride=%d must be a multiple of 4 and >= cout=%d", y_stride, c->cw.cout);
    ConvParamsW q;
    memset(&q, 0, sizeof(q));
    q.seg[0] = ConvSeg{x, rup(c->cin, 4), x_stride};
    q.nseg = 1;
    q.wpk = c->cw.wpk_w; q.bias = c->cw.bias; q.out = y; q.out_stride = y_stride;
    q.cout_store = std::min(rup(c->cw.cout, 4), y_stride);
    q.cout_pad = c->cw.cout_pad;
    q.B = B; q.H = H; q.W = W; q.nchunk = c->cw.nchunk_w; q.lrelu = leaky;
    if (tile == 4) {
#ifdef PIVLFN_TOOLS
        PIV_REQUIRE(c->cw.wpk_w4, "conv2d_wino4: the layer object carries no F(4x4) weights");
        q.wpk = c->cw.wpk_w4; q.nchunk = c->cw.nchunk_w4;
        return launch_conv_w4(q, st);
#else
        PIV_REQUIRE(false, "conv2d_wino: the F(4x4) kernel is part of the tools build only");
#endif
    }
    return launch_conv_w(q, st);
}

// Standalone 3 x 3 / stride 1 / pad 1 layer on the split-operand Winograd kernel (conv_wino_b3.hip): fp32 in, fp32 out; terms = 6, 8 or 9.
int conv_forward_wb(const pivlfn_conv *c, const float *x, int x_stride, float *y, int y_stride, int B, int H, int W, int leaky,
                    int terms, hipStream_t st)
{
    PIV_REQUIRE(c && x && y, "conv2d_wino_b3: null argument");
    PIV_REQUIRE(c->cw.wpk_wb, "conv2d_wino_b3: the layer is not 3 x 3 with whole 64-channel output groups (k=%dx%d, cout_pad=%d)", c->cw.KH, c->cw.KW, c->cw.cout_pad);
    PIV_REQUIRE(x_stride % 4 == 0 && x_stride >= rup(c->cin, 4), "conv2d_wino_b3: x_stride=%d must be a multiple of 4 and >= %d", x_stride, rup(c->cin, 4));
    PIV_REQUIRE(y_stride % 4 == 0 && y_stride >= c->cw.cout, "conv2d_wino_b3: y_stride=%d must be a multiple of 4 and >= cout=%d", y_stride, c->cw.cout);
    ConvParamsW q;
    memset(&q, 0, sizeof(q));
    q.seg[0] = ConvSeg{x, rup(c->cin, 4), x_stride};
    q.nseg = 1;
    q.wpk_b = c->cw.wpk_wb; q.bias = c->cw.bias; q.out = y; q.out_stride = y_stride;
    q.cout_store = std::min(rup(c->cw.cout, 4), y_stride);
    q.cout_pad = c->cw.cout_pad;
    q.B = B; q.H = H; q.W = W; q.nchunk = c->cw.nstep_wb; q.lrelu = leaky; q.terms = terms;
    return launch_conv_wb(q, st);
}

int net_forward(pivlfn_net *net, const float *img1, const float *img2, float *flow, float *levels, int B, int H, int W,
                void *ws, size_t ws_bytes, hipStream_t st)
{
    PIV_REQUIRE(net && img1 && img2 && flow && ws, "forward: null argument");
    PIV_REQUIRE(B > 0 && H >= 32 && W >= 32 && H % 32 == 0 && W % 32 == 0,
                "forward: H=%d W=%d must be positive multiples of 32 (use estimate() for other sizes)", H, W);
    PIV_REQUIRE((reinterpret_cast<size_t>(ws) & 255) == 0, "forward: workspace must be 256-byte aligned");
    t_precision = net->precision == 5 ? 0 : net->precision;
    t_no_b3 = net->precision == 5;
    t_side = net->side;
    Plan pl; Buffers bf;
    pl.base = reinterpret_cast<char *>(ws);
    plan(net, B, H, W, pl, bf);
    if (pl.off > ws_bytes) {
        set_error("forward: workspace of %zu bytes is too small, need %zu", ws_bytes, pl.off);
        return PIVLFN_ERR_WORKSPACE;
    }
    int h[7], w[7];
    for (int L = 1; L <= 6; ++L) { h[L] = H >> (L - 1); w[L] = W >> (L - 1); }
    const int N2 = 2 * B;
    t_scratch = bf.ksplit;
#define RUN(expr) do { int _rc = (expr); if (_rc) return _rc; } while (0)
    // mean subtraction + layout change (:321-323), image pyramid (:336-343)
    RUN(launch_prep_images(img1, img2, bf.img[1], B, H, W, net->mean, st));
    for (int L = 2; L <= 6; ++L) RUN(launch_resize_nhwc4(bf.img[L - 1], bf.img[L], N2, h[L - 1], w[L - 1], h[L], w[L], st));
    // Flow-independent 1x1 convs on the side stream: NetC_ext (:353-355) for levels <= 2 and Regularization.moduleFeat (:227-232,
    // applied at :280) for levels < 5; 1.8 GB of HBM traffic at 1024^2.  Where they run is a trade: beside NetC they slow its
    // MFMA-bound kernels and push the level-3 features out of the Infinity Cache; beside levels 6-4 those chains of tiny kernels
    // wait for CUs (the level-5 warp+correlation 15 -> 40 us).  Round 2, early: issued per level as soon as NetC had produced the
    // level's features (the level-3 launch gained 2.6 us).  Late round 2: the conv stacks are twice as fast, the prefetch pass below
    // restores the level-3 features whatever the order, and all of it after NetC is the faster step (10.58 vs 10.70 ms,
    // tools/net_ab.py --masks 0,4096, three interleaved rounds) -- the default again; the early order stays selectable in the tools build.
    const hipStream_t side = (PIV_KNOB(1) & 2048) ? st : net->side;        // tools A/B: everything on one stream
    // Whatever path leaves this function -- also an early error return -- the main stream joins every piece of side-stream work
    // that was forked and not yet waited for: an un-joined fork would invalidate a stream capture and let the side kernels run on
    // into the caller's next use of the workspace.
    struct SideJoin {
        hipStream_t st;
        hipEvent_t *ev;
        unsigned pending = 0;
        ~SideJoin()
        {
            for (int L = 0; L < 7; ++L)
                if (pending & (1u << L)) (void)hipStreamWaitEvent(st, ev[L], 0);
        }
    } side_join{st, net->ev_join};
    const bool side_early = (PIV_KNOB(1) & 4096) != 0;                     // tools A/B: 4096 = each level's share right behind its NetC layer
    bool fused1 = false;          // level 1's two 1 x 1 layers were computed inside NetC.conv1's kernel (below)
    auto side_level = [&](int L) -> int {
        if (L < net->lowest || L > 4 || (L == 1 && fused1)) return PIVLFN_OK;
        if (side != st) {
            PIV_CHECK_HIP(hipEventRecord(net->ev_fork[L], st));
            PIV_CHECK_HIP(hipStreamWaitEvent(side, net->ev_fork[L], 0));
        }
        RUN(conv(net->lv[L].feat, {{bf.feat[L], C_FEAT[L], C_FEAT[L]}}, bf.featR[L], 128, 128, nullptr, 0, 1, B, h[L], w[L], 1, 0, 0, side));
        if (L <= 2)
            RUN(conv(net->ext[L], {{bf.feat[L], 32, 32}}, bf.ext[L], 64, 64, nullptr, 0, 1, N2, h[L], w[L], 1, 0, 0, side));
        PIV_CHECK_HIP(hipEventRecord(net->ev_join[L], side));
        if (side != st) side_join.pending |= 1u << L;
        return PIVLFN_OK;
    };
    // NetC on both frames as one batch of 2B (:325-326, Features.forward :108-116)
    const ConvW *nc = net->netc;
    // Level 1's NetC_ext (32 -> 64, both frames) and moduleFeat (32 -> 128, first frame) read nothing but conv1's output: in the fp32
    // modes they are computed from conv1's activated accumulators in its own kernel -- their 1 GB of writes goes out under conv1's
    // matrix work instead of beside the latency-bound chains of levels 6-4, and conv1's output is not read back twice.
    if (net->lowest == 1 && net->fuse1_w && (t_precision == 0 || t_precision == 4) && !(PIV_KNOB(1) & 536870912)) {
        ConvParams p1;
        memset(&p1, 0, sizeof(p1));
        p1.seg[0] = ConvSeg{bf.img[1], 4, 4}; p1.nseg = 1;
        p1.wpk = nc[0].wpk; p1.bias = nc[0].bias; p1.out = bf.feat[1]; p1.out_stride = 32; p1.cout_store = 32; p1.cout_pad = nc[0].cout_pad;
        p1.B = N2; p1.H = h[1]; p1.W = w[1]; p1.Ho = h[1]; p1.Wo = w[1];
        p1.KH = 7; p1.KW = 7; p1.S = 1; p1.padY = 3; p1.padX = 3;
        p1.nchunk = nc[0].nchunk; p1.tail = nc[0].tail; p1.lrelu = 1; p1.cin_real = nc[0].cin;
        const Conv1Fuse f1{net->fuse1_w, net->fuse1_b, bf.ext[1], bf.featR[1], B};
        const int rc1 = launch_conv1_fused(p1, f1, st);
        if (rc1 > 0) return rc1;
        fused1 = rc1 == 0;
    }
    if (!fused1) RUN(conv(nc[0], {{bf.img[1], 4, 4}}, bf.feat[1], 32, 32, nullptr, 0, 1, N2, h[1], w[1], 1, 3, 3, st));
    if (side_early) RUN(side_level(1));
    RUN(conv(nc[1], {{bf.feat[1], 32, 32}}, bf.sa, 32, 32, nullptr, 0, 1, N2, h[1], w[1], 2, 1, 1, st));
    RUN(conv(nc[2], {{bf.sa, 32, 32}}, bf.sb, 32, 32, nullptr, 0, 1, N2, h[2], w[2], 1, 1, 1, st));
    RUN(conv(nc[3], {{bf.sb, 32, 32}}, bf.feat[2], 32, 32, nullptr, 0, 1, N2, h[2], w[2], 1, 1, 1, st));
    if (side_early) RUN(side_level(2));
    RUN(conv(nc[4], {{bf.feat[2], 32, 32}}, bf.sa, 64, 64, nullptr, 0, 1, N2, h[2], w[2], 2, 1, 1, st));
    RUN(conv(nc[5], {{bf.sa, 64, 64}}, bf.feat[3], 64, 64, nullptr, 0, 1, N2, h[3], w[3], 1, 1, 1, st));
    if (side_early) RUN(side_level(3));
    RUN(conv(nc[6], {{bf.feat[3], 64, 64}}, bf.sa, 96, 96, nullptr, 0, 1, N2, h[3], w[3], 2, 1, 1, st));
    RUN(conv(nc[7], {{bf.sa, 96, 96}}, bf.feat[4], 96, 96, nullptr, 0, 1, N2, h[4], w[4], 1, 1, 1, st));
    if (side_early) RUN(side_level(4));
    RUN(conv(nc[8], {{bf.feat[4], 96, 96}}, bf.feat[5], 128, 128, nullptr, 0, 1, N2, h[4], w[4], 2, 1, 1, st));
    RUN(conv(nc[9], {{bf.feat[5], 128, 128}}, bf.feat[6], 192, 192, nullptr, 0, 1, N2, h[5], w[5], 2, 1, 1, st));
    if (!side_early)
        for (int L = 4; L >= net->lowest; --L) RUN(side_level(L));
    // The side stream's 1x1 outputs (1.5 GB at 1024^2) pass through the Infinity Cache after NetC wrote the level-3 features and
    // push them out; the level-3 warp+correlation -- one tile per CU, nothing to overlap a miss with -- then gathers from HBM.  A
    // read-only pass over those features at the tail of the side stream (it runs beside levels 6-4, which use a fraction of the
    // chip) brings them back.  Nothing depends on it; it is joined at the end of the forward.
    bool touched = false;
    if (side != st && net->lowest <= 3 && !(PIV_KNOB(1) & 16384)) {
        RUN(launch_touch(bf.feat[3], (size_t)N2 * h[3] * w[3] * C_FEAT[3], bf.mean, side));
        PIV_CHECK_HIP(hipEventRecord(net->ev_join[6], side));      // joined at the very end of the forward (stream capture needs it)
        side_join.pending |= 1u << 6;
        touched = true;
    }

    float *prev = nullptr, *cur = bf.flowA;
    size_t lvoff = 0;
    for (int L = 6; L >= net->lowest; --L) {
        const LevelW &lw = net->lv[L];
        const int hh = h[L], ww = w[L], k = K_LEVEL[L], cm = C_MATCH[L], cf = C_FEAT[L];
        const size_t half = (size_t)B * hh * ww;
        const float *f1m = L <= 2 ? bf.ext[L] : bf.feat[L];
        const float *f2m = f1m + half * cm;
        const float *f1raw = bf.feat[L];
        const float *im1 = bf.img[L], *im2 = bf.img[L] + half * 4;
        const float sc = net->scale[L];
        const int s = L >= 4 ? 1 : 2;
        // fp16 mode: the hidden activations of this level's conv stacks are stored as fp16 (same size test as conv())
        const int h16 = (t_precision == 1 && (long)hh * ww >= 64 * 64) ? 1 : 0;
        // Join the side stream only where its results are first read: NetC_ext feeds Matching at levels <= 2, moduleFeat feeds
        // Regularization at levels 3 and 4.  (A cross-queue wait costs a barrier packet and a cold start for the next
        // kernel: in front of the level-3 warp+correlation it cost that launch 2 us.)
        if (L <= 2 && !(L == 1 && fused1)) { PIV_CHECK_HIP(hipStreamWaitEvent(st, net->ev_join[L], 0)); side_join.pending &= ~(1u << L); }
        // ---- Matching (:165-187)
        const float *fup = nullptr;
        if (prev) {
            RUN(launch_dwconvT(prev, lw.upconv, bf.flow_up, B, h[L + 1], w[L + 1], 2, 4, 4, 4, st));
            fup = bf.flow_up;
        }
        const bool prof = net->prof_level == L && net->ev_used + 4 <= net->ev.size();
        if (net->prof_level == L && !prof) net->ev_dropped++;
        if (prof) {
            // (a) start/stop events attached to the dispatch itself; (b) a plain event pair around it (reported for reference)
            warp_corr_time_next(net->ev[net->ev_used], net->ev[net->ev_used + 1]);
            PIV_CHECK_HIP(hipEventRecord(net->ev[net->ev_used + 2], st));
        }
        RUN(launch_warp_corr(f1m, f2m, fup, sc, bf.corr, B, cm, hh, ww, s, 1, true, st));
        if (prof) {
            PIV_CHECK_HIP(hipEventRecord(net->ev[net->ev_used + 3], st));
            net->ev_used += 4;
        }
        const float *cin = bf.corr;
        if (s == 2) {
            RUN(launch_dwconvT(bf.corr, lw.upcorr, bf.corr_up, B, hh / 2, ww / 2, 49, 56, 56, 56, st));
            cin = bf.corr_up;
        }
        const float *hid = nullptr;       // output of the last hidden layer (32 channels)
        {
            const float *src = cin;
            int cprev = 0;
            for (int j = 0; j < net->nstack; ++j) {
                const int wd = net->width[j];
                float *dst = (j & 1) ? bf.t128b : bf.t128a;
                const int o16 = (j + 1 < net->nstack) ? h16 : 0;      // the flow head reads fp32
                if (j == 0) RUN(conv(lw.M[0], {{src, 52, 56}}, dst, wd, wd, nullptr, 0, 1, B, hh, ww, 1, 1, 1, st, 0, o16));
                else RUN(conv(lw.M[j], {{src, cprev, cprev}}, dst, wd, wd, nullptr, 0, 1, B, hh, ww, 1, 1, 1, st, h16, o16));
                src = dst;
                cprev = wd;
            }
            hid = src;
        }
        if (PIV_KNOB(1) & 1)    // A/B: heads on the matrix cores (30 of 32 output columns wasted)
            RUN(conv(lw.M[net->nstack], {{hid, 32, 32}}, bf.flowM, 4, 4, fup, 4, 0, B, hh, ww, 1, k / 2, k / 2, st));
        else
            RUN(launch_conv_head(hid, lw.headM, lw.hbM[0], lw.hbM[1], fup, bf.flowM, B, hh, ww, k, st));
        // ---- Subpixel (:209-217)
        RUN(launch_backwarp_nhwc(f2m, bf.flowM, sc, bf.f2w, B, hh, ww, cm, st));
        {
            const float *src = nullptr;
            int cprev = 0;
            for (int j = 0; j < net->nstack; ++j) {
                const int wd = net->width[j];
                float *dst = (j & 1) ? bf.t128b : bf.t128a;
                const int o16 = (j + 1 < net->nstack) ? h16 : 0;
                if (j == 0) RUN(conv(lw.S[0], {{f1m, cm, cm}, {bf.f2w, cm, cm}, {bf.flowM, 4, 4}}, dst, wd, wd, nullptr, 0, 1, B, hh, ww, 1, 1, 1, st, 0, o16));
                else RUN(conv(lw.S[j], {{src, cprev, cprev}}, dst, wd, wd, nullptr, 0, 1, B, hh, ww, 1, 1, 1, st, h16, o16));
                src = dst;
                cprev = wd;
            }
            hid = src;
        }
        if (PIV_KNOB(1) & 1)
            RUN(conv(lw.S[net->nstack], {{hid, 32, 32}}, bf.flowS, 4, 4, bf.flowM, 4, 0, B, hh, ww, 1, k / 2, k / 2, st));
        else
            RUN(launch_conv_head(hid, lw.headS, lw.hbS[0], lw.hbS[1], bf.flowM, bf.flowS, B, hh, ww, k, st));
        // ---- Regularization (:274-303); note it takes the RAW NetC feature (:361)
        RUN(launch_flow_mean(bf.flowS, bf.partial, nullptr, B, hh * ww, st));       // partial sums only: reg_prep finishes the mean
        RUN(launch_reg_prep(im1, im2, bf.flowS, bf.mean, bf.partial, sc, bf.misc4, B, hh, ww, st));
        if (L == 3 || L == 4) { PIV_CHECK_HIP(hipStreamWaitEvent(st, net->ev_join[L], 0)); side_join.pending &= ~(1u << L); }
        const float *fr = L < 5 ? bf.featR[L] : f1raw;
        const int cfr = L < 5 ? 128 : cf;
        RUN(conv(lw.R[0], {{fr, cfr, cfr}, {bf.misc4, 4, 4}}, bf.t128a, 128, 128, nullptr, 0, 1, B, hh, ww, 1, 1, 1, st, 0, h16));
        RUN(conv(lw.R[1], {{bf.t128a, 128, 128}}, bf.t128b, 128, 128, nullptr, 0, 1, B, hh, ww, 1, 1, 1, st, h16, h16));
        RUN(conv(lw.R[2], {{bf.t128b, 128, 128}}, bf.t64a, 64, 64, nullptr, 0, 1, B, hh, ww, 1, 1, 1, st, h16, h16));
        RUN(conv(lw.R[3], {{bf.t64a, 64, 64}}, bf.t64b, 64, 64, nullptr, 0, 1, B, hh, ww, 1, 1, 1, st, h16, h16));
        RUN(conv(lw.R[4], {{bf.t64b, 64, 64}}, bf.t32a, 32, 32, nullptr, 0, 1, B, hh, ww, 1, 1, 1, st, h16, h16));
        RUN(conv(lw.R[5], {{bf.t32a, 32, 32}}, bf.t32b, 32, 32, nullptr, 0, 1, B, hh, ww, 1, 1, 1, st, h16, h16));
        const int kk = k * k, kkp = rup(kk, 4);
        if (L < 5) {     // (k x 1) then (1 x k), no activation in between (:253-261); d1 and dist stay fp32
            RUN(conv(lw.dist0, {{bf.t32b, 32, 32}}, bf.d1, kkp, kkp, nullptr, 0, 0, B, hh, ww, 1, k / 2, 0, st, h16, 0));
            RUN(conv(lw.dist1, {{bf.d1, kkp, kkp}}, bf.dist, kkp, kkp, nullptr, 0, 0, B, hh, ww, 1, 0, k / 2, st));
        } else {
            RUN(conv(lw.dist0, {{bf.t32b, 32, 32}}, bf.dist, kkp, kkp, nullptr, 0, 0, B, hh, ww, 1, k / 2, k / 2, st, h16, 0));
        }
        const bool last = L == net->lowest;
        RUN(launch_reg_tail(bf.dist, kkp, bf.flowS, lw.wx, lw.wy, lw.bx, lw.by, k, cur, last ? flow : nullptr,
                            net->scale[1], B, hh, ww, st));
        if (levels) {
            RUN(launch_flow4_to_nchw(bf.flowM, levels + lvoff, B, hh, ww, st)); lvoff += half * 2;
            RUN(launch_flow4_to_nchw(bf.flowS, levels + lvoff, B, hh, ww, st)); lvoff += half * 2;
            RUN(launch_flow4_to_nchw(cur, levels + lvoff, B, hh, ww, st)); lvoff += half * 2;
        }
        prev = cur;
        cur = (cur == bf.flowA) ? bf.flowB : bf.flowA;
    }
    if (touched) { PIV_CHECK_HIP(hipStreamWaitEvent(st, net->ev_join[6], 0)); side_join.pending &= ~(1u << 6); }
#undef RUN
    return PIVLFN_OK;
}

}  // namespace pivlfn
