// a4r_encoder_layer_fwd / a4r_encoder_layer_bwd (ABI 409; SURVEY 8(b)'s `encoder_layer_fwd / bwd`): ONE C call per post-LN encoder layer with the
// reference's serial Houlsby wrappers on both sub-layers (HF BertLayer under Downstream/Text/model/encoders.py:39-56; BertAdaptedSelfOutput,
// model/model.py:292-297, injected at run.py:452-465), frozen backbone.  Host code only: the call enqueues, on the given stream, exactly the
// launches adapter4rec_amd/engine.py's _block_forward / _block_backward issue for such a layer -- same entry points, same arguments, same
// order -- so the results are bit-identical to the per-kernel path (tests/test_layer_call_gpu.py).  What it saves is the host side: 7 + 9 ctypes
// calls with ~25 arguments each become 2 with one descriptor (tools/cpu_enqueue.py; profiles/r06_c_launch_gaps.txt shows the GPU is never
// waiting for the host at the benchmarked sizes, so this is a boundary, not a speed-up).
#include <hip/hip_runtime.h>

#include "../../include/a4r.h"

namespace {

bool ok_h(int H) { return H == 128 || H == 256 || H == 512 || H == 768; }

int check(const a4r_encoder_layer_t* l) {
    if (!l || l->M <= 0 || l->M % 128 || !ok_h(l->H) || l->F <= 0 || l->F % 64 || l->S <= 0 || l->S > 32 || l->n_items <= 0 || l->n_heads <= 0) return A4R_EINVAL;
    if (l->n_heads * l->dh != l->H || (!l->offsets && (long)l->n_items * l->S > l->M)) return A4R_EINVAL;      // (packed items: M counts their own tokens)
    if (!l->wqkv || !l->wo || !l->wi || !l->wo2 || !l->bqkv || !l->bo || !l->bi || !l->bo2 || !l->ln1_g || !l->ln1_b || !l->ln2_g || !l->ln2_b) return A4R_EINVAL;
    if (!l->qkv || !l->ctx || !l->h1 || !l->zp1 || !l->z1 || !l->u || !l->upre || !l->h2 || !l->zp2 || !l->z2 || !l->st1 || !l->st2) return A4R_EINVAL;
    for (const auto& a : l->ad) {
        if (!a.wd || !a.wu || !a.bd || !a.bu) return A4R_EINVAL;
        const int nf = (a.wd_f != nullptr) + (a.wu_f != nullptr) + (a.wdT_f != nullptr) + (a.wuT_f != nullptr);
        if (nf != 0 && nf != 4) return A4R_EINVAL;
        const int ng = (a.g_wu != nullptr) + (a.g_wd != nullptr) + (a.g_bu != nullptr) + (a.g_bd != nullptr);
        if (ng != 0 && ng != 4) return A4R_EINVAL;               // an adapter is trainable as a whole (weights and biases ride in one a4r_gemm_tn2 launch) or frozen
    }
    return A4R_OK;
}

a4r_gemm_t gemm(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K) {
    a4r_gemm_t g{};
    g.A = A; g.B = B; g.C = C; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.in_dtype = A4R_BF16; g.out_dtype = A4R_BF16; g.act = A4R_ACT_NONE; g.dact = A4R_ACT_NONE; g.alpha = 1.f;
    return g;
}

a4r_attn_t attn(const a4r_encoder_layer_t* l) {
    a4r_attn_t a{};
    a.qkv = l->qkv; a.ld = 3 * l->H; a.q_off = 0; a.k_off = l->H; a.v_off = 2 * l->H;
    a.key_mask = l->key_mask; a.offsets = l->offsets;
    a.n_items = l->n_items; a.S = l->S; a.n_heads = l->n_heads; a.dh = l->dh; a.causal = l->causal; a.dtype = A4R_BF16;
    a.scale = l->scale; a.mask_neg = l->mask_neg;
    a.drop_p = l->p_attn; a.drop_site = l->drop_site; a.drop_seed = l->drop_seed;
    return a;
}

// dense -> dropout -> adapter -> LayerNorm(residual + .) of one half (engine.py: _sub_forward, the one-launch serial form)
int sub_forward(void* s, const a4r_encoder_layer_t* l, int half, const void* dense_in, int K, const void* w, const float* bias, const void* resid,
                const void* resid_lo, const float* ln_g, const float* ln_b, void* h, void* zp, void* z, void* v, void* out, void* out_lo, float* st) {
    const int M = l->M, H = l->H;
    const a4r_layer_adapter_t& a = l->ad[half];
    a4r_gemm_t g = gemm(dense_in, K, w, K, h, H, M, H, K);
    g.bias = bias; g.drop_p = l->p_hidden; g.drop_site = l->drop_site + 1 + half; g.drop_seed = l->drop_seed;
    if (int rc = a4r_gemm_nt(s, &g)) return rc;
    const bool frag = a.wd_f != nullptr, lo = resid_lo != nullptr || out_lo != nullptr;
    const int ldlo = l->lo_nibble ? H / 2 : H;                   // bytes per row of a plane
    return a4r_adapter_ln_fwd(s, h, H, h, H, resid, H, frag ? a.wd_f : a.wd, a.bd, frag ? a.wu_f : a.wu, a.bu, ln_g, ln_b, l->ln_eps, a.act, zp, z,
                              v, v ? H : 0, out, H, st, M, H, 64, A4R_BF16, nullptr, 0, nullptr,
                              reinterpret_cast<const float*>(lo ? resid_lo : nullptr), lo && resid_lo ? ldlo : 0,
                              reinterpret_cast<float*>(lo ? out_lo : nullptr), lo && out_lo ? ldlo : 0, (frag ? 1 : 0) | (lo ? (l->lo_nibble ? 6 : 2) : 0));
}

// backward of sub_forward (engine.py: _sub_backward, fused form with both bias gradients in the weight-gradient launch): dy -> dh (gradient of the
// dense output, through its dropout mask), dv (gradient of the residual input)
int sub_backward(void* s, const a4r_encoder_layer_t* l, int half, const void* dy, const void* v_or_y, bool is_y, const float* st, const float* ln_g,
                 const float* ln_b, const void* zp, const void* z, const void* h, void* dv) {
    const int M = l->M, H = l->H;
    const a4r_layer_adapter_t& a = l->ad[half];
    const bool frag = a.wuT_f != nullptr;
    if (int rc = a4r_adapter_ln_bwd(s, dy, H, v_or_y, H, st, ln_g, nullptr, 0, zp, a.act, frag ? a.wuT_f : a.wuT, frag ? a.wdT_f : a.wdT, 1, dv, H, l->dzp, l->d_h, H,
                                    nullptr, nullptr, nullptr, M, H, 64, A4R_BF16, l->p_hidden, l->drop_site + 1 + half, l->drop_seed, nullptr, frag ? 2 : 0,
                                    is_y ? ln_b : nullptr))
        return rc;
    if (!a.g_wu) return A4R_OK;
    // dW_up = dv^T z, dW_down = dzp^T h and both bias gradients (column sums of dv / dzp) from one launch
    return a4r_gemm_tn2(s, dv, H, z, 64, a.g_wu, a.ldg_wu, H, 64, l->dzp, 64, h, H, a.g_wd, a.ldg_wd, 64, H, M, A4R_BF16, a.g_bu, a.g_bd);
}

}  // namespace

extern "C" int a4r_encoder_layer_fwd(void* stream, const a4r_encoder_layer_t* l, const void* x, void* x1, void* x_out) {
    if (int rc = check(l)) return rc;
    if (!x || !x1 || !x_out) return A4R_EINVAL;
    const int M = l->M, H = l->H, F = l->F;
    a4r_gemm_t g = gemm(x, H, l->wqkv, H, l->qkv, 3 * H, M, 3 * H, H);
    g.bias = l->bqkv;
    if (int rc = a4r_gemm_nt(stream, &g)) return rc;
    a4r_attn_t a = attn(l);
    a.out = l->ctx; a.ldo = H;
    if (int rc = a4r_attn_fwd(stream, &a)) return rc;
    if (int rc = sub_forward(stream, l, 0, l->ctx, H, l->wo, l->bo, x, l->x_lo, l->ln1_g, l->ln1_b, l->h1, l->zp1, l->z1, l->v1, x1, l->x1_lo, l->st1)) return rc;
    g = gemm(x1, H, l->wi, H, l->u, F, M, F, H);
    g.bias = l->bi; g.act = A4R_ACT_GELU; g.C2 = l->upre; g.ldc2 = F; g.c2_mode = l->upre_q8 ? 2 : 1; g.q8_tiled = l->q8_tiled;
    if (int rc = a4r_gemm_nt(stream, &g)) return rc;
    return sub_forward(stream, l, 1, l->u, F, l->wo2, l->bo2, x1, l->x1_lo, l->ln2_g, l->ln2_b, l->h2, l->zp2, l->z2, l->v2, x_out, l->xout_lo, l->st2);
}

extern "C" int a4r_encoder_layer_bwd(void* stream, const a4r_encoder_layer_t* l, const void* x1, const void* x_out, const void* dx_out, void* dx_in) {
    if (int rc = check(l)) return rc;
    if (!x1 || !x_out || !dx_out || !l->wqkvT || !l->woT || !l->wiT || !l->wo2T || !l->dv1 || !l->dv2 || !l->dzp || !l->d_h || !l->du || !l->dx1 || !l->dctx || !l->dqkv)
        return A4R_EINVAL;
    for (const auto& a : l->ad)
        if (!a.wdT || !a.wuT) return A4R_EINVAL;
    const int M = l->M, H = l->H, F = l->F;
    // FFN half: LayerNorm + adapter backward, weight gradients, d FFN-down * gelu', d FFN-up + the residual branch
    if (int rc = sub_backward(stream, l, 1, dx_out, l->v2 ? l->v2 : x_out, l->v2 == nullptr, l->st2, l->ln2_g, l->ln2_b, l->zp2, l->z2, l->h2, l->dv2)) return rc;
    a4r_gemm_t g = gemm(l->d_h, H, l->wo2T, H, l->du, F, M, F, H);
    g.Pre = l->upre; g.ldpre = F; g.dact = l->upre_q8 ? A4R_DACT_MUL_Q8 : A4R_DACT_MUL; g.q8_tiled = l->q8_tiled;
    if (int rc = a4r_gemm_nt(stream, &g)) return rc;
    g = gemm(l->du, F, l->wiT, F, l->dx1, H, M, H, F);
    g.R1 = l->dv2; g.ldr1 = H;
    if (int rc = a4r_gemm_nt(stream, &g)) return rc;
    // attention half
    if (int rc = sub_backward(stream, l, 0, l->dx1, l->v1 ? l->v1 : x1, l->v1 == nullptr, l->st1, l->ln1_g, l->ln1_b, l->zp1, l->z1, l->h1, l->dv1)) return rc;
    if (!dx_in) return A4R_OK;                                   // first layer of a frozen tower: nothing trainable below its attention
    g = gemm(l->d_h, H, l->woT, H, l->dctx, H, M, H, H);
    if (int rc = a4r_gemm_nt(stream, &g)) return rc;
    a4r_attn_t a = attn(l);
    a.dout = l->dctx; a.ldo = H; a.dqkv = l->dqkv;
    if (int rc = a4r_attn_bwd(stream, &a)) return rc;
    g = gemm(l->dqkv, 3 * H, l->wqkvT, 3 * H, dx_in, H, M, H, 3 * H);
    g.R1 = l->dv1; g.ldr1 = H;
    return a4r_gemm_nt(stream, &g);
}
