// Implicit-GEMM convolution forward / data-gradient on MFMA (gfx950): the C-ABI entry points and the fp32 / bf16 kernel instances.
// Kernels, launchers and the dispatch rules are templates in conv_fprop_kernels.h; the f16 forward-operand instances are compiled by
// conv_fprop_f16.hip.
#include <string.h>

#define SA_FPROP_MAIN_TU
#include "conv_fprop_kernels.h"

// ncls = 0: one geometry (sa_conv_fprop); ncls >= 2: geoms[0 .. ncls) differ only in in_off / out_off (checked), wpks[c] = class c's packed operand
static int conv_fprop_impl(const sa_conv_geom* g, int ncls, int dtype, const void* in, const void* const* wpks, void* out, const sa_epilogue* ep, void* stream) {
    using namespace sa;
    const void* wpk = wpks[0];
    const int vec = dtype == SA_F32 ? 4 : 8;
    const int bke = dtype == SA_F32 ? 32 : 64;
    if (dtype != SA_F32 && dtype != SA_BF16 && dtype != SA_F16) return SA_EUNSUPPORTED;
    // f16 is a FORWARD operand type: 16-bit addends share its type or are bf16 by add_dtype; masks (data gradients) are never f16
    if (ep->mask && ep->mask_mode != SA_MASK_NONE && ep->mask_dtype == SA_F16) return SA_EUNSUPPORTED;
    if (dtype == SA_F16 ? (ep->out_dtype == SA_BF16 || (ep->addend && ep->add_dtype == SA_BF16)) : (ep->out_dtype == SA_F16 || (ep->addend && ep->add_dtype == SA_F16)))
        return SA_EUNSUPPORTED;   // the 16-bit addend / output of a launch share its operand type (a bf16 copy of an f16 output: ep->out_lp)
    const int ntaps = g->KT[0] * g->KT[1] * g->KT[2];
    if (g->Cin % vec || g->Kpad % bke || g->Kpad < ntaps * g->Cin || g->CoutPad % 128 || g->cout_valid > g->CoutPad ||
        g->cout_valid > g->Cout || ntaps < 1 || ntaps > SA_MAX_TAPS)
        return SA_EINVAL;
    const int64_t M = (int64_t)g->N * g->Dm * g->Hm * g->Wm;
    if (M <= 0 || M >= (1ll << 31)) return SA_EINVAL;
    FpropArgs a;
    a.in = in;
    a.wpk = wpk;
    a.out = out;
    a.ep = *ep;
    a.g = *g;
    a.dW = make_fastdiv(g->Wm);
    a.dH = make_fastdiv(g->Hm);
    a.dD = make_fastdiv(g->Dm);
    a.dTw = make_fastdiv(g->KT[2]);
    a.dThw = make_fastdiv(g->KT[1] * g->KT[2]);
    a.dCv = make_fastdiv(g->Cin / vec);
    a.M = (uint32_t)M;
    a.ntaps = ntaps;
    a.nk = g->Kpad / bke;
    a.nblk_m = (uint32_t)((M + 127) / 128);
    a.dCin = make_fastdiv(g->Cin);
    a.w2pk = nullptr;
    a.bias1 = nullptr;
    a.h_out = nullptr;
    a.dbg = g_tunables.pp_dbg;
    a.group_m = 0;
    a.ncls = 0;
    for (int c = 0; c < 8; ++c) {
        a.cls_wpk[c] = nullptr;
        for (int d = 0; d < 3; ++d) a.cls_in_off[c][d] = a.cls_out_off[c][d] = 0;
    }
    if (ncls >= 2) {
        if (ncls > 8) return SA_EUNSUPPORTED;
        for (int c = 0; c < ncls; ++c) {
            sa_conv_geom t = g[c];
            for (int d = 0; d < 3; ++d) {
                a.cls_in_off[c][d] = t.in_off[d];
                a.cls_out_off[c][d] = t.out_off[d];
                t.in_off[d] = g->in_off[d];
                t.out_off[d] = g->out_off[d];
            }
            if (memcmp(&t, g, sizeof t) != 0 || !wpks[c]) return SA_EUNSUPPORTED;      // (anything else differs: the caller launches them one by one)
            a.cls_wpk[c] = wpks[c];
        }
        a.ncls = (uint32_t)ncls;
    }
    {
        const int sz = dtype == SA_F32 ? 4 : 2;
        const uint64_t ib = (uint64_t)g->N * g->Di * g->Hi * g->Wi * g->Cin * sz, wb = (uint64_t)g->CoutPad * g->Kpad * sz;
        const bool fits = ib < 0xfffffff0ull - 4096 && wb < 0xfffffff0ull && !dbg(SA_DBG_NO_DMA);
        a.in_bytes = fits ? (uint32_t)ib : 0u;
        a.w_bytes = fits ? (uint32_t)wb : 0u;
    }
    hipStream_t st = (hipStream_t)stream;
    if (dtype == SA_F16) return a.in_bytes ? dispatch_fprop_f16(a, st) : SA_EUNSUPPORTED;
    return dtype == SA_F32 ? dispatch_fprop<float>(a, st) : dispatch_fprop<bf16_t>(a, st);
}

extern "C" int sa_conv_fprop(const sa_conv_geom* g, int dtype, const void* in, const void* wpk, void* out, const sa_epilogue* ep,
                             void* stream) {
    if (!g || !in || !wpk || !out || !ep) return SA_EINVAL;
    return conv_fprop_impl(g, 0, dtype, in, &wpk, out, ep, stream);
}

// The launch geometries of ONE layer that differ only in in_off / out_off and their packed operands -- the eight output-parity classes of
// nn.ConvTranspose3d k4 s2 p1 (baseline.py:283-293) and of the strided convolution's data gradient -- in one launch (grid = n x the blocks of a class).
// SA_EUNSUPPORTED (nothing launched) when the geometries differ in anything else or the kernel the dispatcher would pick does not take classes: the caller then
// issues sa_conv_fprop per geometry.  Seven launch tails fewer per layer; at the small levels a class is a fraction of a round of the 256 CUs.
extern "C" int sa_conv_fprop_classes(const sa_conv_geom* geoms, int n, int dtype, const void* in, const void* const* wpks, void* out, const sa_epilogue* ep,
                                     void* stream) {
    if (!geoms || n < 1 || !in || !wpks || !wpks[0] || !out || !ep) return SA_EINVAL;
    if (n == 1) return conv_fprop_impl(geoms, 0, dtype, in, wpks, out, ep, stream);
    if (sa::dbg(SA_DBG_NO_CLASS_LAUNCH)) return SA_EUNSUPPORTED;
    return conv_fprop_impl(geoms, n, dtype, in, wpks, out, ep, stream);
}

// Residual block forward in ONE launch (reference src/networks/vqvae/baseline.py:150-160):
//   y = relu(x + conv1x1x1(relu(conv3x3x3(x) + b1)) + b2);  h = relu(conv3x3x3(x) + b1) optionally stored for the backward pass.
extern "C" int sa_resblock_fprop(const sa_conv_geom* g, int dtype, const void* x, const void* w3pk, const float* bias1, const void* w1pk, void* h_out,
                                 void* y_out, const sa_epilogue* ep, void* stream) {
    using namespace sa;
    if (!g || !x || !w3pk || !bias1 || !w1pk || !y_out || !ep) return SA_EINVAL;
    if (ep->out_dtype != dtype || (ep->addend && ep->add_dtype != dtype)) return SA_EUNSUPPORTED;
    if ((dtype != SA_BF16 && dtype != SA_F16) || g->cout_valid != 128 || g->Cout != 128 || g->cin_valid != 128 || g->Cin != 128) return SA_EUNSUPPORTED;
    const int ntaps = g->KT[0] * g->KT[1] * g->KT[2];
    if (g->Kpad % 64 || g->Kpad < ntaps * g->Cin || g->CoutPad != 128 || ntaps < 1 || ntaps > SA_MAX_TAPS) return SA_EINVAL;
    // the fused kernel writes h and y at the linear voxel index: identity output map only
    for (int d = 0; d < 3; ++d)
        if (g->out_mult[d] != 1 || g->out_off[d] != 0) return SA_EUNSUPPORTED;
    if (g->Dm != g->Do || g->Hm != g->Ho || g->Wm != g->Wo) return SA_EUNSUPPORTED;
    const int64_t M = (int64_t)g->N * g->Dm * g->Hm * g->Wm;
    if (M <= 0 || M >= (1ll << 31)) return SA_EINVAL;
    const uint64_t ib = (uint64_t)g->N * g->Di * g->Hi * g->Wi * g->Cin * 2, wb = (uint64_t)g->CoutPad * g->Kpad * 2;
    if (ib >= 0xfffffff0ull - 4096) return SA_EUNSUPPORTED;
    FpropArgs a;
    a.in = x;
    a.wpk = w3pk;
    a.out = y_out;
    a.ep = *ep;
    a.g = *g;
    a.dW = make_fastdiv(g->Wm);
    a.dH = make_fastdiv(g->Hm);
    a.dD = make_fastdiv(g->Dm);
    a.dTw = make_fastdiv(g->KT[2]);
    a.dThw = make_fastdiv(g->KT[1] * g->KT[2]);
    a.dCv = make_fastdiv(g->Cin / 8);
    a.dCin = make_fastdiv(g->Cin);
    a.M = (uint32_t)M;
    a.ntaps = ntaps;
    a.nk = g->Kpad / 64;
    a.nblk_m = (uint32_t)((M + 127) / 128);
    a.in_bytes = (uint32_t)ib;
    a.w_bytes = (uint32_t)wb;
    a.w2pk = w1pk;
    a.bias1 = bias1;
    a.h_out = h_out;
    a.dbg = 0;
    a.group_m = 0;
    a.ncls = 0;
    return dtype == SA_F16 ? launch_resblock_f16(a, (hipStream_t)stream) : launch_resblock<bf16_t>(a, (hipStream_t)stream);
}
