// TORCH_LIBRARY op layer over the C ABI ("a PyTorch C++/HIP extension registering ops, each taking / returning at::Tensor on
// the current device ... TORCH_CHECK -> Python RuntimeError ... kernels enqueue on c10::hip::getCurrentHIPStream()").
//
// The product's Python host (adapter4rec_amd/engine.py) drives the kernels through ctypes (adapter4rec_amd/_lib.py): nothing there needs
// torch's dispatcher.  This file is the same boundary in the form a C++ / TorchScript / torch.compile caller binds: namespace `a4r`, one op per
// hot-path kernel family, arguments are tensors (shape, dtype, device and contiguity checked here; strides taken from the tensors), outputs
// are caller-allocated (ownership stays with the caching allocator), nothing is retained past the call, no allocation, no synchronisation.
// No compute lives here: every op forwards to ONE extern "C" entry point of liba4r_hip.so (include/a4r.h) and raises on its status code.
//
//   torch.ops.a4r.gemm_nt                  a4r_gemm_nt            nn.Linear forward / dgrad + epilogue      (model/encoders.py:53 -> HF BertLayer)
//   torch.ops.a4r.adapter_residual_ln_fwd  a4r_adapter_ln_fwd     BertAdaptedSelfOutput.forward             (model/model.py:292-297)
//   torch.ops.a4r.adapter_residual_ln_bwd  a4r_adapter_ln_bwd     its backward
//   torch.ops.a4r.ln_fwd                   a4r_ln_fwd             LayerNorm (+ position add, dropout)       (HF BertSelfOutput / modules.py:57-66)
//   torch.ops.a4r.score_bce_fwd / _bwd     a4r_score_bce_*        Model.forward / ModelCPC.forward head     (model/model.py:58-68,127-133)
//   torch.ops.a4r.fused_adam_step          a4r_adam_step          optim.Adam over the flat buffers, lr groups (run.py:505-529)
//   torch.ops.a4r.topk_rank_eval           a4r_eval_rank          eval_model's per-user rank                (data_utils/metrics.py:82-116)
//   torch.ops.a4r.lora_bwd                 a4r_lora_bwd_fused     loralib Linear (query, value): every low-rank gradient in one pass (run_adapter.py:384-395)
//   torch.ops.a4r.encoder_layer_fwd / _bwd a4r_encoder_layer_fwd / _bwd  one post-LN HF BertLayer + serial Houlsby wrappers per call (model/encoders.py:39-56, model/model.py:292-297)
//   torch.ops.a4r.sasrec_block_fwd / _bwd  a4r_sasrec_block_fwd / _bwd   one adapted SASRec TransformerBlock per call (model/modules.py:45-87, model/model.py:341-376)
//   torch.ops.a4r.embed_ln_fwd             a4r_embed_ln           HF BertEmbeddings / RobertaEmbeddings (word + position + type -> LayerNorm -> dropout)
//   torch.ops.a4r.patch_embed_fwd          a4r_patchify           ViTPatchEmbeddings' im2col (+ the uint8 ToTensor / Normalize of dataset.py:77-81); the projection is gemm_nt
//   torch.ops.a4r.vit_assemble             a4r_vit_assemble       cls token + position rows around the projected patches (HF ViTEmbeddings / ViTMAEEmbeddings)
//   torch.ops.a4r.abi_version              a4r_version
// (SURVEY 8(b) also lists lora_qv_fwd and allreduce_flat: a LoRA projection's forward IS gemm_nt on the merged operand (a4r_lora_merge, once per step; its
//  backward is lora_bwd above), and the gradient exchange is torch.distributed's all_reduce on the flat buffer (adapter4rec_amd/ddp.py) -- RCCL has no
//  C entry point in this library to wrap.)
#include <ATen/ATen.h>
#include <c10/hip/HIPStream.h>
#include <torch/library.h>

#include "../../include/a4r.h"

namespace {

using at::Tensor;
using c10::optional;

void* cur_stream(const Tensor& t) { return static_cast<void*>(c10::hip::getCurrentHIPStream(t.device().index()).stream()); }

int dt_of(const Tensor& t) {
    if (t.scalar_type() == at::kBFloat16) return A4R_BF16;
    if (t.scalar_type() == at::kFloat) return A4R_F32;
    if (t.scalar_type() == at::kByte) return A4R_FP8;           // OCP e4m3 bit patterns
    TORCH_CHECK(false, "a4r: unsupported dtype ", t.scalar_type(), " (bf16, fp32 or uint8 = e4m3 bits)");
}

void chk_mat(const Tensor& t, const char* name, const Tensor& like) {
    TORCH_CHECK(t.defined(), "a4r: ", name, " is undefined");
    TORCH_CHECK(t.is_cuda(), "a4r: ", name, " must be a device tensor (no CPU path)");
    TORCH_CHECK(t.device() == like.device(), "a4r: ", name, " is on ", t.device(), ", expected ", like.device());
    TORCH_CHECK(t.dim() == 2 && t.stride(1) == 1, "a4r: ", name, " must be 2-D with unit column stride, got sizes ", t.sizes(), " strides ", t.strides());
}
void chk_f32_vec(const optional<Tensor>& t, const char* name, const Tensor& like, int64_t n) {
    if (!t.has_value() || !t->defined()) return;
    TORCH_CHECK(t->is_cuda() && t->device() == like.device() && t->scalar_type() == at::kFloat && t->is_contiguous() && t->numel() >= n,
                "a4r: ", name, " must be a contiguous fp32 device tensor of >= ", n, " elements");
}
const void* cptr(const optional<Tensor>& t) { return (t.has_value() && t->defined()) ? t->data_ptr() : nullptr; }
void* mptr(const optional<Tensor>& t) { return (t.has_value() && t->defined()) ? t->data_ptr() : nullptr; }
int ld_of(const optional<Tensor>& t) { return (t.has_value() && t->defined()) ? (int)t->stride(0) : 0; }
void status(int rc, const char* what) {
    TORCH_CHECK(rc == 0, "a4r: ", what, " returned ", rc, rc == -1 ? " (invalid argument: shape / alignment / dtype combination not supported)" : " (launch failure)");
}

// C = epilogue(alpha * A B^T): A [M, K], B [N, K] (nn.Linear weight layout), C [M, N]; act: A4R_ACT_*; optional bias [N] fp32, residuals
// R1 / R2 [M, N] (C's dtype), counter-based dropout (drop_p, site, seed; drop_first = before the residuals, BertSelfOutput order).
void gemm_nt(const Tensor& A, const Tensor& B, Tensor C, const optional<Tensor>& bias, const optional<Tensor>& R1, const optional<Tensor>& R2,
             int64_t act, double alpha, double drop_p, int64_t drop_site, int64_t drop_seed, bool drop_first) {
    chk_mat(A, "A", A); chk_mat(B, "B", A); chk_mat(C, "C", A);
    TORCH_CHECK(A.size(1) == B.size(1), "a4r::gemm_nt: A is [M, K] = ", A.sizes(), ", B must be [N, K], got ", B.sizes());
    TORCH_CHECK(C.size(0) == A.size(0) && C.size(1) == B.size(0), "a4r::gemm_nt: C must be [", A.size(0), ", ", B.size(0), "], got ", C.sizes());
    TORCH_CHECK(A.scalar_type() == B.scalar_type(), "a4r::gemm_nt: A and B must share a dtype");
    chk_f32_vec(bias, "bias", A, B.size(0));
    a4r_gemm_t g{};
    g.A = A.data_ptr(); g.B = B.data_ptr(); g.C = C.data_ptr(); g.bias = static_cast<const float*>(cptr(bias));
    for (const auto* r : {&R1, &R2})
        if (r->has_value() && (*r)->defined()) {
            chk_mat(**r, "residual", A);
            TORCH_CHECK((*r)->sizes() == C.sizes() && (*r)->scalar_type() == C.scalar_type(), "a4r::gemm_nt: a residual must match C in shape and dtype");
        }
    g.R1 = cptr(R1); g.R2 = cptr(R2); g.ldr1 = ld_of(R1); g.ldr2 = ld_of(R2);
    g.M = (int)A.size(0); g.N = (int)B.size(0); g.K = (int)A.size(1);
    g.lda = (int)A.stride(0); g.ldb = (int)B.stride(0); g.ldc = (int)C.stride(0);
    g.in_dtype = dt_of(A); g.out_dtype = dt_of(C);
    TORCH_CHECK(g.in_dtype != A4R_FP8, "a4r::gemm_nt: e4m3 operands need per-row scales -- use the C ABI (a4r_gemm_t.scale_a / scale_b)");
    g.act = (int)act; g.dact = A4R_ACT_NONE; g.alpha = (float)alpha;
    g.drop_p = (float)drop_p; g.drop_site = (uint32_t)drop_site; g.drop_seed = (uint64_t)drop_seed; g.drop_first = drop_first ? 1 : 0;
    status(a4r_gemm_nt(cur_stream(A), &g), "a4r_gemm_nt");
}

// zp = A Wd^T + bd; z = act(zp); v = z Wu^T + bu + R1 + R2; y = LayerNorm(v)   (A must be R1 or R2, or R2 absent: include/a4r.h)
void adapter_residual_ln_fwd(const Tensor& A, const Tensor& R1, const optional<Tensor>& R2, const Tensor& Wd, const Tensor& bd, const Tensor& Wu,
                             const Tensor& bu, const Tensor& gamma, const Tensor& beta, double eps, int64_t act, Tensor zp, Tensor z,
                             const optional<Tensor>& v, Tensor y, Tensor stats, const optional<Tensor>& res32, const optional<Tensor>& y32) {
    chk_mat(A, "A", A); chk_mat(R1, "R1", A); chk_mat(Wd, "Wd", A); chk_mat(Wu, "Wu", A); chk_mat(zp, "zp", A); chk_mat(z, "z", A); chk_mat(y, "y", A);
    const int64_t M = A.size(0), H = A.size(1), d = Wd.size(0);
    TORCH_CHECK(A.scalar_type() == at::kBFloat16, "a4r::adapter_residual_ln_fwd: bf16 activations only (the fp32 path uses gemm_nt + ln_fwd)");
    TORCH_CHECK(Wd.size(1) == H && Wu.size(0) == H && Wu.size(1) == d, "a4r::adapter_residual_ln_fwd: Wd must be [d, H], Wu [H, d]; got ", Wd.sizes(), " and ", Wu.sizes());
    TORCH_CHECK(zp.size(0) >= M && zp.size(1) == d && z.sizes() == zp.sizes() && y.size(0) >= M && y.size(1) == H, "a4r::adapter_residual_ln_fwd: output shapes");
    chk_f32_vec(bd, "bd", A, d); chk_f32_vec(bu, "bu", A, H); chk_f32_vec(gamma, "gamma", A, H); chk_f32_vec(beta, "beta", A, H);
    TORCH_CHECK(stats.is_cuda() && stats.scalar_type() == at::kFloat && stats.is_contiguous() && stats.numel() >= 2 * M, "a4r::adapter_residual_ln_fwd: stats must be fp32 [M, 2]");
    if (R2.has_value() && R2->defined()) chk_mat(*R2, "R2", A);
    if (v.has_value() && v->defined()) chk_mat(*v, "v", A);
    for (const auto* t : {&res32, &y32})
        if (t->has_value() && (*t)->defined())
            TORCH_CHECK((*t)->is_cuda() && (*t)->scalar_type() == at::kFloat && (*t)->dim() == 2 && (*t)->stride(1) == 1 && (*t)->size(1) == H && (*t)->size(0) >= M,
                        "a4r::adapter_residual_ln_fwd: res32 / y32 must be fp32 [>= M, H]");
    status(a4r_adapter_ln_fwd(cur_stream(A), A.data_ptr(), (int)A.stride(0), R1.data_ptr(), (int)R1.stride(0), cptr(R2), ld_of(R2), Wd.data_ptr(),
                              bd.data_ptr<float>(), Wu.data_ptr(), bu.data_ptr<float>(), gamma.data_ptr<float>(), beta.data_ptr<float>(), (float)eps,
                              (int)act, zp.data_ptr(), z.data_ptr(), mptr(v), ld_of(v), y.data_ptr(), (int)y.stride(0), stats.data_ptr<float>(),
                              (int)M, (int)H, (int)d, A4R_BF16, nullptr, 0, nullptr, static_cast<const float*>(cptr(res32)), ld_of(res32),
                              static_cast<float*>(mptr(y32)), ld_of(y32), /* w_frag: row-major weights through this layer */ 0),
           "a4r_adapter_ln_fwd");
}

// dv = LayerNorm'(dy; v, stats, gamma) [+ dres]; dzp = (dv Wu) * act'(zp); dh = mask * (dzp Wd [+ dv]); optional column sums (include/a4r.h)
void adapter_residual_ln_bwd(const Tensor& dy, const Tensor& v, const Tensor& stats, const Tensor& gamma, const optional<Tensor>& dres, const Tensor& zp,
                             int64_t act, const Tensor& WuT, const Tensor& WdT, bool inner_res, Tensor dv, Tensor dzp, Tensor dh,
                             const optional<Tensor>& dgamma, const optional<Tensor>& dbeta, const optional<Tensor>& dbias, const optional<Tensor>& dbd,
                             double drop_p, int64_t drop_site, int64_t drop_seed, bool bias_total, const optional<Tensor>& beta_y) {
    chk_mat(dy, "dy", dy); chk_mat(v, "v", dy); chk_mat(zp, "zp", dy); chk_mat(WuT, "WuT", dy); chk_mat(WdT, "WdT", dy);
    chk_mat(dv, "dv", dy); chk_mat(dzp, "dzp", dy); chk_mat(dh, "dh", dy);
    const int64_t M = dy.size(0), H = dy.size(1), d = zp.size(1);
    TORCH_CHECK(dy.scalar_type() == at::kBFloat16, "a4r::adapter_residual_ln_bwd: bf16 activations only");
    TORCH_CHECK(WuT.size(0) == d && WuT.size(1) == H && WdT.size(0) == H && WdT.size(1) == d, "a4r::adapter_residual_ln_bwd: WuT must be [d, H], WdT [H, d]");
    chk_f32_vec(gamma, "gamma", dy, H); chk_f32_vec(dgamma, "dgamma", dy, H); chk_f32_vec(dbeta, "dbeta", dy, H); chk_f32_vec(dbias, "dbias", dy, H);
    chk_f32_vec(dbd, "dbd", dy, d); chk_f32_vec(beta_y, "beta_y", dy, H);
    TORCH_CHECK(stats.is_cuda() && stats.scalar_type() == at::kFloat && stats.is_contiguous() && stats.numel() >= 2 * M, "a4r::adapter_residual_ln_bwd: stats must be fp32 [M, 2]");
    if (dres.has_value() && dres->defined()) chk_mat(*dres, "dres", dy);
    status(a4r_adapter_ln_bwd(cur_stream(dy), dy.data_ptr(), (int)dy.stride(0), v.data_ptr(), (int)v.stride(0), stats.data_ptr<float>(), gamma.data_ptr<float>(),
                              cptr(dres), ld_of(dres), zp.data_ptr(), (int)act, WuT.data_ptr(), WdT.data_ptr(), inner_res ? 1 : 0, dv.data_ptr(), (int)dv.stride(0),
                              dzp.data_ptr(), dh.data_ptr(), (int)dh.stride(0), static_cast<float*>(mptr(dgamma)), static_cast<float*>(mptr(dbeta)),
                              static_cast<float*>(mptr(dbias)), (int)M, (int)H, (int)d, A4R_BF16, (float)drop_p, (uint32_t)drop_site, (uint64_t)drop_seed,
                              static_cast<float*>(mptr(dbd)), bias_total ? 1 : 0, static_cast<const float*>(cptr(beta_y))),
           "a4r_adapter_ln_bwd");
}

// y = LayerNorm(v [+ add[row % add_rows]]) (+ dropout); stats [M, 2] = (mean, rstd)
void ln_fwd(const Tensor& v, const optional<Tensor>& add, const Tensor& gamma, const Tensor& beta, double eps, Tensor y, Tensor stats,
            double drop_p, int64_t drop_site, int64_t drop_seed) {
    chk_mat(v, "v", v); chk_mat(y, "y", v);
    const int64_t M = v.size(0), H = v.size(1);
    TORCH_CHECK(y.sizes() == v.sizes() && y.scalar_type() == v.scalar_type(), "a4r::ln_fwd: y must match v");
    chk_f32_vec(gamma, "gamma", v, H); chk_f32_vec(beta, "beta", v, H);
    int add_rows = 0;
    if (add.has_value() && add->defined()) {
        TORCH_CHECK(add->is_cuda() && add->scalar_type() == at::kFloat && add->is_contiguous() && add->dim() == 2 && add->size(1) == H, "a4r::ln_fwd: add must be fp32 [rows, H]");
        add_rows = (int)add->size(0);
    }
    TORCH_CHECK(stats.is_cuda() && stats.scalar_type() == at::kFloat && stats.is_contiguous() && stats.numel() >= 2 * M, "a4r::ln_fwd: stats must be fp32 [M, 2]");
    status(a4r_ln_fwd(cur_stream(v), v.data_ptr(), (int)v.stride(0), static_cast<const float*>(cptr(add)), add_rows, gamma.data_ptr<float>(), beta.data_ptr<float>(),
                      (float)eps, y.data_ptr(), (int)y.stride(0), stats.data_ptr<float>(), (int)M, (int)H, dt_of(v), (float)drop_p, (uint32_t)drop_site,
                      (uint64_t)drop_seed),
           "a4r_ln_fwd");
}

void chk_f32(const Tensor& t, const char* name, const Tensor& like, int64_t n) {
    TORCH_CHECK(t.is_cuda() && t.device() == like.device() && t.scalar_type() == at::kFloat && t.is_contiguous() && t.numel() >= n,
                "a4r: ", name, " must be a contiguous fp32 device tensor of >= ", n, " elements, got ", t.sizes(), " ", t.scalar_type());
}

// emb [B, L, 2, E] (positive | negative item embeddings), prec [B, L-1, E], log_mask [B, L-1] -> pos, neg [B, L-1], loss_ws [4]
void score_bce_fwd(const Tensor& emb, const Tensor& prec, const Tensor& log_mask, Tensor pos, Tensor neg, Tensor loss_ws, int64_t B, int64_t L, int64_t E, bool cpc) {
    chk_f32(emb, "emb", emb, B * L * 2 * E); chk_f32(prec, "prec", emb, B * (L - 1) * E); chk_f32(log_mask, "log_mask", emb, B * (L - 1));
    chk_f32(pos, "pos", emb, B * (L - 1)); chk_f32(neg, "neg", emb, B * (L - 1)); chk_f32(loss_ws, "loss_ws", emb, 4);
    status(a4r_score_bce_fwd(cur_stream(emb), emb.data_ptr<float>(), prec.data_ptr<float>(), log_mask.data_ptr<float>(), pos.data_ptr<float>(),
                             neg.data_ptr<float>(), loss_ws.data_ptr<float>(), (int)B, (int)L, (int)E, cpc ? 1 : 0),
           "a4r_score_bce_fwd");
}
void score_bce_bwd(const Tensor& emb, const Tensor& prec, const Tensor& log_mask, const Tensor& pos, const Tensor& neg, const Tensor& loss_ws, double loss_scale,
                   const optional<Tensor>& loss_scale_dev, Tensor d_prec, Tensor d_emb, int64_t B, int64_t L, int64_t E, bool cpc) {
    chk_f32(emb, "emb", emb, B * L * 2 * E); chk_f32(prec, "prec", emb, B * (L - 1) * E); chk_f32(log_mask, "log_mask", emb, B * (L - 1));
    chk_f32(pos, "pos", emb, B * (L - 1)); chk_f32(neg, "neg", emb, B * (L - 1)); chk_f32(loss_ws, "loss_ws", emb, 4);
    chk_f32(d_prec, "d_prec", emb, B * (L - 1) * E); chk_f32(d_emb, "d_emb", emb, B * L * 2 * E);
    chk_f32_vec(loss_scale_dev, "loss_scale_dev", emb, 1);
    status(a4r_score_bce_bwd(cur_stream(emb), emb.data_ptr<float>(), prec.data_ptr<float>(), log_mask.data_ptr<float>(), pos.data_ptr<float>(), neg.data_ptr<float>(),
                             loss_ws.data_ptr<float>(), (float)loss_scale, static_cast<const float*>(cptr(loss_scale_dev)), d_prec.data_ptr<float>(),
                             d_emb.data_ptr<float>(), (int)B, (int)L, (int)E, cpc ? 1 : 0),
           "a4r_score_bce_bwd");
}

// torch.optim.Adam over flat fp32 buffers; segment i covers [seg_end[i-1], seg_end[i]) and uses group_lr[seg_group[i]]
void fused_adam_step(Tensor p, const Tensor& g, Tensor m, Tensor v, const Tensor& seg_end, const Tensor& seg_group, const Tensor& group_lr, int64_t step,
                     double beta1, double beta2, double eps, double grad_scale) {
    const int64_t n = p.numel();
    chk_f32(p, "p", p, n); chk_f32(g, "g", p, n); chk_f32(m, "m", p, n); chk_f32(v, "v", p, n);
    TORCH_CHECK(seg_end.is_cuda() && seg_end.scalar_type() == at::kInt && seg_end.is_contiguous() && seg_group.is_cuda() && seg_group.scalar_type() == at::kInt &&
                    seg_group.is_contiguous() && seg_group.numel() == seg_end.numel(),
                "a4r::fused_adam_step: seg_end / seg_group must be contiguous int32 device tensors of equal length");
    chk_f32(group_lr, "group_lr", p, 1);
    TORCH_CHECK(step >= 1, "a4r::fused_adam_step: step counts from 1");
    status(a4r_adam_step(cur_stream(p), p.data_ptr<float>(), g.data_ptr<float>(), m.data_ptr<float>(), v.data_ptr<float>(), n, seg_end.data_ptr<int32_t>(),
                         seg_group.data_ptr<int32_t>(), (int)seg_end.numel(), group_lr.data_ptr<float>(), (int)step, (float)beta1, (float)beta2, (float)eps,
                         (float)grad_scale),
           "a4r_adam_step");
}

// rank[u] = 1 + #{items i != target[u], i not in history(u), i >= 1 : score(u, i) > score(u, target[u])}  (metrics.py:82-116)
void topk_rank_eval(const Tensor& prec, const Tensor& item_emb, const Tensor& target, const Tensor& hist_ptr, const Tensor& hist_idx, Tensor rank) {
    TORCH_CHECK(prec.dim() == 2 && item_emb.dim() == 2 && prec.size(1) == item_emb.size(1), "a4r::topk_rank_eval: prec [U, E], item_emb [N + 1, E]");
    const int64_t U = prec.size(0), N1 = item_emb.size(0), E = prec.size(1);
    chk_f32(prec, "prec", prec, U * E); chk_f32(item_emb, "item_emb", prec, N1 * E);
    for (const Tensor* t : {&target, &hist_ptr, &hist_idx, (const Tensor*)&rank})
        TORCH_CHECK(t->is_cuda() && t->scalar_type() == at::kInt && t->is_contiguous(), "a4r::topk_rank_eval: target / hist_ptr / hist_idx / rank must be contiguous int32 device tensors");
    TORCH_CHECK(target.numel() == U && rank.numel() == U && hist_ptr.numel() == U + 1, "a4r::topk_rank_eval: target [U], rank [U], hist_ptr [U + 1] (CSR)");
    status(a4r_eval_rank(cur_stream(prec), prec.data_ptr<float>(), item_emb.data_ptr<float>(), target.data_ptr<int32_t>(), hist_ptr.data_ptr<int32_t>(),
                         hist_idx.data_ptr<int32_t>(), rank.data_ptr<int32_t>(), (int)U, (int)N1, (int)E),
           "a4r_eval_rank");
}

// The low-rank gradients of a block's two LoRAs (query: a, value: b) from one pass over x [M, H] and the two slices dqa, dqb [M, H] of the fused qkv
// gradient.  Aa, Ab, BTa (= B_a^T), BTb: [R, H] views (R = 8: ranks <= 8, or 16: ranks <= 15; rows past the rank zero); dAa, dAb [R, H] and dBa, dBb
// [H, >= R] fp32, ACCUMULATED into (dB without the LoRA scaling); dbias_a / dbias_b: optional strided fp32 vectors += column sums of dqa / dqb.
void lora_bwd(const Tensor& x, const Tensor& dqa, const Tensor& dqb, const Tensor& Aa, const Tensor& Ab, const Tensor& BTa, const Tensor& BTb, double scale_a,
              double scale_b, Tensor dAa, Tensor dAb, Tensor dBa, Tensor dBb, const optional<Tensor>& dbias_a, const optional<Tensor>& dbias_b, Tensor ws) {
    chk_mat(x, "x", x); chk_mat(dqa, "dqa", x); chk_mat(dqb, "dqb", x);
    const int64_t M = x.size(0), H = x.size(1), R = Aa.size(0);
    TORCH_CHECK(dqa.sizes() == x.sizes() && dqb.sizes() == x.sizes() && dqa.stride(0) == dqb.stride(0), "a4r::lora_bwd: dqa / dqb must be [M, H] views with one row stride");
    for (const Tensor* w : {&Aa, &Ab, &BTa, &BTb}) {
        chk_mat(*w, "weight operand", x);
        TORCH_CHECK(w->scalar_type() == x.scalar_type() && w->size(0) == R && w->size(1) == H && w->stride(0) == Aa.stride(0),
                    "a4r::lora_bwd: Aa, Ab, BTa, BTb must be [R, H] views of x's dtype with one row stride");
    }
    TORCH_CHECK(R == 8 || R == 16, "a4r::lora_bwd: 8 (ranks <= 8) or 16 (ranks <= 15) rank rows, got ", R);
    for (const Tensor* o : {(const Tensor*)&dAa, (const Tensor*)&dAb, (const Tensor*)&dBa, (const Tensor*)&dBb}) {
        chk_mat(*o, "gradient", x);
        TORCH_CHECK(o->scalar_type() == at::kFloat, "a4r::lora_bwd: gradients are fp32");
    }
    TORCH_CHECK(dAa.size(0) == R && dAb.size(0) == R && dAa.size(1) == H && dAb.size(1) == H && dAa.stride(0) == dAb.stride(0), "a4r::lora_bwd: dAa, dAb must be [R, H]");
    TORCH_CHECK(dBa.size(0) == H && dBb.size(0) == H && dBa.stride(0) == dBb.stride(0), "a4r::lora_bwd: dBa, dBb must be [H, >= R]");
    int ldbias = 0;
    for (const optional<Tensor>* b : {&dbias_a, &dbias_b})
        if (b->has_value() && (*b)->defined()) {
            TORCH_CHECK((*b)->is_cuda() && (*b)->scalar_type() == at::kFloat && (*b)->dim() == 1 && (*b)->numel() == H, "a4r::lora_bwd: bias sums are fp32 [H] (strided) device vectors");
            ldbias = (int)(*b)->stride(0);
        }
    chk_f32(ws, "ws", x, a4r_lora_bwd_fused_ws_floats((int)H));
    status(a4r_lora_bwd_fused(cur_stream(x), x.data_ptr(), (int)x.stride(0), dqa.data_ptr(), dqb.data_ptr(), (int)dqa.stride(0), Aa.data_ptr(), Ab.data_ptr(),
                              BTa.data_ptr(), BTb.data_ptr(), (int)Aa.stride(0), (float)scale_a, (float)scale_b, dAa.data_ptr<float>(), dAb.data_ptr<float>(),
                              (int)dAa.stride(0), dBa.data_ptr<float>(), dBb.data_ptr<float>(), (int)dBa.stride(0), (float*)mptr(dbias_a), (float*)mptr(dbias_b), ldbias,
                              (int)M, (int)H, dt_of(x), (int)R, ws.data_ptr<float>(), ws.numel()),
           "a4r_lora_bwd_fused");
}

void chk_dev(const Tensor& t, const char* name, const Tensor& like, at::ScalarType st) {
    TORCH_CHECK(t.defined() && t.is_cuda() && t.device() == like.device(), "a4r: ", name, " must be a device tensor on ", like.device(), " (no CPU path)");
    TORCH_CHECK(t.scalar_type() == st, "a4r: ", name, " must be ", st, ", got ", t.scalar_type());
    TORCH_CHECK(t.dim() == 0 || t.stride(t.dim() - 1) == 1, "a4r: ", name, " must have unit stride in its last dimension");
}
void chk_list(at::TensorList l, size_t n, const char* name) { TORCH_CHECK(l.size() == n, "a4r: ", name, " must hold ", n, " tensors, got ", l.size()); }

// One post-LN encoder layer with serial Houlsby adapters on both halves (include/a4r.h: a4r_encoder_layer_t).  Tensor lists, in the struct's order:
//   w     = [wqkv, bqkv, wo, bo, wi, bi, wo2, bo2, ln1_g, ln1_b, ln2_g, ln2_b]     (weights bf16 [out, in], the rest fp32)
//   ad1/2 = [wd [64, H], bd [64], wu [H, 64], bu [H]]                               (row-major through this layer)
//   saved = [qkv, ctx, h1, v1, zp1, z1, u, upre, h2, v2, zp2, z2, st1, st2]          (upre bf16 [M, F], or uint8 = the 8-bit derivative; v1 / v2 always kept here)
void fill_layer(a4r_encoder_layer_t& l, const Tensor& x, at::TensorList w, at::TensorList ad1, at::TensorList ad2, at::TensorList saved, const optional<Tensor>& key_mask,
                const optional<Tensor>& offsets, int64_t n_items, int64_t S, int64_t n_heads, bool causal, double scale, double mask_neg, double ln_eps, double p_attn,
                double p_hidden, int64_t drop_site, int64_t drop_seed, int64_t act1, int64_t act2, bool q8_tiled) {
    chk_mat(x, "x", x);
    TORCH_CHECK(x.scalar_type() == at::kBFloat16, "a4r::encoder_layer: bf16 activations only");
    chk_list(w, 12, "w"); chk_list(ad1, 4, "ad1"); chk_list(ad2, 4, "ad2"); chk_list(saved, 14, "saved");
    const int64_t M = x.size(0), H = x.size(1), F = w[4].size(0);
    for (int i : {0, 2, 4, 6}) chk_dev(w[i], "a weight matrix", x, at::kBFloat16);
    for (int i : {1, 3, 5, 7, 8, 9, 10, 11}) chk_dev(w[i], "a bias / LayerNorm vector", x, at::kFloat);
    TORCH_CHECK(w[0].size(0) == 3 * H && w[0].size(1) == H && w[2].size(0) == H && w[2].size(1) == H && w[4].size(1) == H && w[6].size(0) == H && w[6].size(1) == F,
                "a4r::encoder_layer: weight shapes must be [3H, H], [H, H], [F, H], [H, F]");
    l.M = (int)M; l.H = (int)H; l.F = (int)F; l.n_items = (int)n_items; l.S = (int)S; l.n_heads = (int)n_heads; l.dh = (int)(H / (n_heads > 0 ? n_heads : 1)); l.causal = causal;
    l.scale = (float)scale; l.mask_neg = (float)mask_neg; l.ln_eps = (float)ln_eps; l.p_attn = (float)p_attn; l.p_hidden = (float)p_hidden;
    l.drop_site = (uint32_t)drop_site; l.drop_seed = (uint64_t)drop_seed;
    l.key_mask = static_cast<const float*>(cptr(key_mask)); l.offsets = static_cast<const int32_t*>(cptr(offsets));
    l.wqkv = w[0].data_ptr(); l.bqkv = w[1].data_ptr<float>(); l.wo = w[2].data_ptr(); l.bo = w[3].data_ptr<float>(); l.wi = w[4].data_ptr(); l.bi = w[5].data_ptr<float>();
    l.wo2 = w[6].data_ptr(); l.bo2 = w[7].data_ptr<float>(); l.ln1_g = w[8].data_ptr<float>(); l.ln1_b = w[9].data_ptr<float>(); l.ln2_g = w[10].data_ptr<float>(); l.ln2_b = w[11].data_ptr<float>();
    int k = 0;
    for (at::TensorList a : {ad1, ad2}) {
        chk_dev(a[0], "wd", x, at::kBFloat16); chk_dev(a[1], "bd", x, at::kFloat); chk_dev(a[2], "wu", x, at::kBFloat16); chk_dev(a[3], "bu", x, at::kFloat);
        TORCH_CHECK(a[0].size(0) == 64 && a[0].size(1) == H && a[2].size(0) == H && a[2].size(1) == 64, "a4r::encoder_layer: adapter matrices must be [64, H] and [H, 64]");
        l.ad[k].wd = a[0].data_ptr(); l.ad[k].bd = a[1].data_ptr<float>(); l.ad[k].wu = a[2].data_ptr(); l.ad[k].bu = a[3].data_ptr<float>();
        l.ad[k].act = (int)(k == 0 ? act1 : act2);
        ++k;
    }
    const int64_t cols[14] = {3 * H, H, H, H, 64, 64, F, F, H, H, 64, 64, 2, 2};
    for (int i = 0; i < 14; ++i) {
        const bool f32 = i >= 12, any8 = i == 7;
        TORCH_CHECK(saved[i].is_cuda() && saved[i].dim() == 2 && saved[i].size(0) >= M && saved[i].size(1) == cols[i] && saved[i].is_contiguous(),
                    "a4r::encoder_layer: saved[", i, "] must be a contiguous device tensor [>= M, ", cols[i], "]");
        TORCH_CHECK(f32 ? saved[i].scalar_type() == at::kFloat : (saved[i].scalar_type() == at::kBFloat16 || (any8 && saved[i].scalar_type() == at::kByte)), "a4r::encoder_layer: saved[", i, "] dtype");
    }
    l.qkv = saved[0].data_ptr(); l.ctx = saved[1].data_ptr(); l.h1 = saved[2].data_ptr(); l.v1 = saved[3].data_ptr(); l.zp1 = saved[4].data_ptr(); l.z1 = saved[5].data_ptr();
    l.u = saved[6].data_ptr(); l.upre = saved[7].data_ptr(); l.h2 = saved[8].data_ptr(); l.v2 = saved[9].data_ptr(); l.zp2 = saved[10].data_ptr(); l.z2 = saved[11].data_ptr();
    l.st1 = saved[12].data_ptr<float>(); l.st2 = saved[13].data_ptr<float>();
    l.upre_q8 = saved[7].scalar_type() == at::kByte; l.q8_tiled = q8_tiled && l.upre_q8;
}

void encoder_layer_fwd(const Tensor& x, at::TensorList w, at::TensorList ad1, at::TensorList ad2, at::TensorList saved, Tensor x1, Tensor x_out, const optional<Tensor>& key_mask,
                       const optional<Tensor>& offsets, int64_t n_items, int64_t S, int64_t n_heads, bool causal, double scale, double mask_neg, double ln_eps, double p_attn,
                       double p_hidden, int64_t drop_site, int64_t drop_seed, int64_t act1, int64_t act2, bool q8_tiled) {
    a4r_encoder_layer_t l{};
    fill_layer(l, x, w, ad1, ad2, saved, key_mask, offsets, n_items, S, n_heads, causal, scale, mask_neg, ln_eps, p_attn, p_hidden, drop_site, drop_seed, act1, act2, q8_tiled);
    chk_mat(x1, "x1", x); chk_mat(x_out, "x_out", x);
    TORCH_CHECK(x1.sizes() == x.sizes() && x_out.sizes() == x.sizes() && x1.is_contiguous() && x_out.is_contiguous() && x.is_contiguous(), "a4r::encoder_layer_fwd: x, x1, x_out must be contiguous [M, H]");
    status(a4r_encoder_layer_fwd(cur_stream(x), &l, x.data_ptr(), x1.data_ptr(), x_out.data_ptr()), "a4r_encoder_layer_fwd");
}

// wT = [wqkvT [H, 3H], woT [H, H], wiT [H, F], wo2T [F, H]]; adT1/2 = [wdT [H, 64], wuT [64, H]]; scratch = [dv1, dv2, dzp, d_h, du, dx1, dctx, dqkv];
// grads1/2 = [g_wu [H, 64], g_wd [64, H], g_bu [H], g_bd [64]] (fp32, accumulated) or empty lists for a frozen adapter
void encoder_layer_bwd(const Tensor& dx_out, const Tensor& x1, const Tensor& x_out, at::TensorList w, at::TensorList wT, at::TensorList ad1, at::TensorList ad2, at::TensorList adT1,
                       at::TensorList adT2, at::TensorList saved, at::TensorList scratch, at::TensorList grads1, at::TensorList grads2, const optional<Tensor>& dx_in,
                       const optional<Tensor>& key_mask, const optional<Tensor>& offsets, int64_t n_items, int64_t S, int64_t n_heads, bool causal, double scale, double mask_neg,
                       double ln_eps, double p_attn, double p_hidden, int64_t drop_site, int64_t drop_seed, int64_t act1, int64_t act2, bool q8_tiled) {
    a4r_encoder_layer_t l{};
    fill_layer(l, x1, w, ad1, ad2, saved, key_mask, offsets, n_items, S, n_heads, causal, scale, mask_neg, ln_eps, p_attn, p_hidden, drop_site, drop_seed, act1, act2, q8_tiled);
    chk_list(wT, 4, "wT"); chk_list(adT1, 2, "adT1"); chk_list(adT2, 2, "adT2"); chk_list(scratch, 8, "scratch");
    for (const auto& t : wT) chk_dev(t, "a transposed weight", x1, at::kBFloat16);
    l.wqkvT = wT[0].data_ptr(); l.woT = wT[1].data_ptr(); l.wiT = wT[2].data_ptr(); l.wo2T = wT[3].data_ptr();
    int k = 0;
    for (at::TensorList a : {adT1, adT2}) {
        chk_dev(a[0], "wdT", x1, at::kBFloat16); chk_dev(a[1], "wuT", x1, at::kBFloat16);
        l.ad[k].wdT = a[0].data_ptr(); l.ad[k].wuT = a[1].data_ptr();
        ++k;
    }
    k = 0;
    for (at::TensorList g : {grads1, grads2}) {
        TORCH_CHECK(g.size() == 0 || g.size() == 4, "a4r::encoder_layer_bwd: grads must be [g_wu, g_wd, g_bu, g_bd] or empty");
        if (g.size() == 4) {
            for (const auto& t : g) chk_dev(t, "an adapter gradient", x1, at::kFloat);
            l.ad[k].g_wu = g[0].data_ptr<float>(); l.ad[k].g_wd = g[1].data_ptr<float>(); l.ad[k].g_bu = g[2].data_ptr<float>(); l.ad[k].g_bd = g[3].data_ptr<float>();
            l.ad[k].ldg_wu = (int)g[0].stride(0); l.ad[k].ldg_wd = (int)g[1].stride(0);
        }
        ++k;
    }
    const int64_t M = l.M, H = l.H, F = l.F, cols[8] = {H, H, 64, H, F, H, H, 3 * H};
    void** dst[8] = {&l.dv1, &l.dv2, &l.dzp, &l.d_h, &l.du, &l.dx1, &l.dctx, &l.dqkv};
    for (int i = 0; i < 8; ++i) {
        chk_dev(scratch[i], "a scratch tensor", x1, at::kBFloat16);
        TORCH_CHECK(scratch[i].dim() == 2 && scratch[i].size(0) >= M && scratch[i].size(1) == cols[i] && scratch[i].is_contiguous(), "a4r::encoder_layer_bwd: scratch[", i, "] must be contiguous [>= M, ", cols[i], "]");
        *dst[i] = scratch[i].data_ptr();
    }
    chk_mat(dx_out, "dx_out", x1);
    status(a4r_encoder_layer_bwd(cur_stream(x1), &l, x1.data_ptr(), x_out.data_ptr(), dx_out.data_ptr(), mptr(dx_in)), "a4r_encoder_layer_bwd");
}

// One adapted SASRec block (include/a4r.h: a4r_sasrec_block_t).  w = [wqkv, wfc, w1, b1, w2, b2, ln1_g, ln1_b, ln2_g, ln2_b, wd1, bd1, wu1, bu1, wd2, bd2, wu2, bu2] (fp32);
// grads (bwd) = [g_wd1, g_bd1, g_wu1, g_bu1, g_wd2, g_bd2, g_wu2, g_bu2] or empty
void fill_sasrec(a4r_sasrec_block_t& b, const Tensor& x, at::TensorList w, int64_t n_heads, int64_t F, int64_t d, int64_t act, bool inner_res, double eps, double mask_neg,
                 double drop_attn, double drop_hidden, int64_t drop_site, int64_t drop_seed) {
    chk_list(w, 18, "w");
    for (const auto& t : w) chk_dev(t, "a block parameter", x, at::kFloat);
    const float** p[18] = {&b.wqkv, &b.wfc, &b.w1, &b.b1, &b.w2, &b.b2, &b.ln1_g, &b.ln1_b, &b.ln2_g, &b.ln2_b, &b.wd1, &b.bd1, &b.wu1, &b.bu1, &b.wd2, &b.bd2, &b.wu2, &b.bu2};
    for (int i = 0; i < 18; ++i) *p[i] = w[i].data_ptr<float>();
    b.E = (int)x.size(-1); b.n_heads = (int)n_heads; b.F = (int)F; b.d = (int)d; b.ldwu = (int)w[12].stride(0); b.act = (int)act; b.inner_res = inner_res;
    b.eps = (float)eps; b.mask_neg = (float)mask_neg; b.drop_attn = (float)drop_attn; b.drop_hidden = (float)drop_hidden; b.drop_site = (uint32_t)drop_site; b.drop_seed = (uint64_t)drop_seed;
}
void sasrec_block_fwd(const Tensor& x, const Tensor& log_mask, Tensor y, at::TensorList w, int64_t n_heads, int64_t F, int64_t d, int64_t act, bool inner_res, double eps,
                      double mask_neg, double drop_attn, double drop_hidden, int64_t drop_site, int64_t drop_seed, bool train) {
    chk_dev(x, "x", x, at::kFloat); chk_dev(log_mask, "log_mask", x, at::kFloat); chk_dev(y, "y", x, at::kFloat);
    TORCH_CHECK(x.dim() == 3 && x.is_contiguous() && y.sizes() == x.sizes() && y.is_contiguous() && log_mask.is_contiguous() && log_mask.numel() == x.size(0) * x.size(1),
                "a4r::sasrec_block_fwd: x, y contiguous [users, T, E]; log_mask [users, T]");
    a4r_sasrec_block_t b{};
    fill_sasrec(b, x, w, n_heads, F, d, act, inner_res, eps, mask_neg, drop_attn, drop_hidden, drop_site, drop_seed);
    status(a4r_sasrec_block_fwd(cur_stream(x), &b, x.data_ptr<float>(), log_mask.data_ptr<float>(), y.data_ptr<float>(), (int)x.size(0), (int)x.size(1), train), "a4r_sasrec_block_fwd");
}
void sasrec_block_bwd(const Tensor& x, const Tensor& log_mask, const Tensor& dy, Tensor dx, at::TensorList w, at::TensorList grads, int64_t n_heads, int64_t F, int64_t d,
                      int64_t act, bool inner_res, double eps, double mask_neg, double drop_attn, double drop_hidden, int64_t drop_site, int64_t drop_seed, bool train) {
    chk_dev(x, "x", x, at::kFloat); chk_dev(log_mask, "log_mask", x, at::kFloat); chk_dev(dy, "dy", x, at::kFloat); chk_dev(dx, "dx", x, at::kFloat);
    TORCH_CHECK(x.dim() == 3 && x.is_contiguous() && dy.sizes() == x.sizes() && dy.is_contiguous() && dx.sizes() == x.sizes() && dx.is_contiguous() && log_mask.is_contiguous(),
                "a4r::sasrec_block_bwd: x, dy, dx contiguous [users, T, E]");
    a4r_sasrec_block_t b{};
    fill_sasrec(b, x, w, n_heads, F, d, act, inner_res, eps, mask_neg, drop_attn, drop_hidden, drop_site, drop_seed);
    TORCH_CHECK(grads.size() == 0 || grads.size() == 8, "a4r::sasrec_block_bwd: grads must be the 8 adapter gradients or empty");
    if (grads.size() == 8) {
        for (const auto& t : grads) chk_dev(t, "an adapter gradient", x, at::kFloat);
        float** g[8] = {&b.g_wd1, &b.g_bd1, &b.g_wu1, &b.g_bu1, &b.g_wd2, &b.g_bd2, &b.g_wu2, &b.g_bu2};
        for (int i = 0; i < 8; ++i) *g[i] = grads[i].data_ptr<float>();
        b.ldg_d = (int)grads[0].stride(0); b.ldg_u = (int)grads[2].stride(0);
    }
    status(a4r_sasrec_block_bwd(cur_stream(x), &b, x.data_ptr<float>(), log_mask.data_ptr<float>(), dy.data_ptr<float>(), dx.data_ptr<float>(), (int)x.size(0), (int)x.size(1), train),
           "a4r_sasrec_block_bwd");
}

// HF BertEmbeddings / RobertaEmbeddings: word[id] + pos + type[0] -> LayerNorm -> dropout; ids int64 [n_items, >= S] (row stride taken from the tensor)
void embed_ln_fwd(const Tensor& ids, const Tensor& word, const Tensor& pos, const Tensor& type0, const Tensor& gamma, const Tensor& beta, double eps, Tensor out, int64_t S,
                  bool roberta, int64_t pad_id, double drop_p, int64_t drop_site, int64_t drop_seed, const optional<Tensor>& pre_out, const optional<Tensor>& stats_out,
                  const optional<Tensor>& key_mask_out) {
    chk_dev(ids, "ids", out, at::kLong); chk_dev(word, "word", out, at::kFloat); chk_dev(pos, "pos", out, at::kFloat); chk_dev(type0, "type0", out, at::kFloat);
    chk_mat(out, "out", out);
    TORCH_CHECK(ids.dim() == 2 && ids.size(1) >= S && word.dim() == 2 && out.size(1) == word.size(1) && out.size(0) >= ids.size(0) * S, "a4r::embed_ln_fwd: ids [n, >= S], word [V, H], out [>= n S, H]");
    chk_f32_vec(gamma, "gamma", out, word.size(1)); chk_f32_vec(beta, "beta", out, word.size(1));
    status(a4r_embed_ln(cur_stream(out), ids.data_ptr<int64_t>(), (int)ids.stride(0), word.data_ptr<float>(), pos.data_ptr<float>(), type0.data_ptr<float>(), gamma.data_ptr<float>(),
                        beta.data_ptr<float>(), (float)eps, out.data_ptr(), (int)out.stride(0), (int)ids.size(0), (int)S, (int)word.size(1), roberta, (int)pad_id, dt_of(out),
                        (float)drop_p, (uint32_t)drop_site, (uint64_t)drop_seed, mptr(pre_out), static_cast<float*>(mptr(stats_out)), static_cast<float*>(mptr(key_mask_out))),
           "a4r_embed_ln");
}

// ViTPatchEmbeddings' im2col: img fp32 [n, C, H, W] (normalised) or uint8 [n, H, W, C] (raw: ToTensor + Normalize(0.5, 0.5) applied here) -> out [n * n_keep, >= C P P]
void patch_embed_fwd(const Tensor& img, Tensor out, int64_t patch, const optional<Tensor>& keep_idx) {
    chk_mat(out, "out", out);
    TORCH_CHECK(img.is_cuda() && img.device() == out.device() && img.dim() == 4 && img.is_contiguous() && (img.scalar_type() == at::kFloat || img.scalar_type() == at::kByte),
                "a4r::patch_embed_fwd: img must be a contiguous device tensor, fp32 [n, C, H, W] or uint8 [n, H, W, C]");
    const bool u8 = img.scalar_type() == at::kByte;
    const int64_t n = img.size(0), Cc = u8 ? img.size(3) : img.size(1), Hi = u8 ? img.size(1) : img.size(2), Wi = u8 ? img.size(2) : img.size(3);
    int64_t n_keep = (Hi / patch) * (Wi / patch);
    if (keep_idx.has_value() && keep_idx->defined()) {
        TORCH_CHECK(keep_idx->is_cuda() && keep_idx->scalar_type() == at::kInt && keep_idx->is_contiguous() && keep_idx->dim() == 2 && keep_idx->size(0) == n, "a4r::patch_embed_fwd: keep_idx must be int32 [n, n_keep]");
        n_keep = keep_idx->size(1);
    }
    TORCH_CHECK(out.size(0) >= n * n_keep && out.size(1) >= Cc * patch * patch, "a4r::patch_embed_fwd: out must be [>= n n_keep, >= C P P]");
    status(a4r_patchify(cur_stream(out), img.data_ptr(), u8 ? 1 : 0, out.data_ptr(), (int)out.stride(0), static_cast<const int32_t*>(cptr(keep_idx)), (int)n_keep, (int)n, (int)Cc, (int)Hi,
                        (int)Wi, (int)patch, dt_of(out)),
           "a4r_patchify");
}
void vit_assemble(const Tensor& patches, const Tensor& cls, const Tensor& pos, Tensor out, int64_t n_items, int64_t n_keep, const optional<Tensor>& keep_idx, int64_t tokens_out) {
    chk_mat(patches, "patches", out); chk_mat(out, "out", out); chk_dev(cls, "cls", out, at::kFloat); chk_dev(pos, "pos", out, at::kFloat);
    status(a4r_vit_assemble(cur_stream(out), patches.data_ptr(), (int)patches.stride(0), cls.data_ptr<float>(), pos.data_ptr<float>(), static_cast<const int32_t*>(cptr(keep_idx)),
                            out.data_ptr(), (int)out.stride(0), (int)n_items, (int)n_keep, (int)cls.numel(), dt_of(out), (int)tokens_out),
           "a4r_vit_assemble");
}

int64_t abi_version() { return a4r_version(); }

}  // namespace

TORCH_LIBRARY(a4r, m) {
    m.def("gemm_nt(Tensor A, Tensor B, Tensor(a!) C, Tensor? bias=None, Tensor? R1=None, Tensor? R2=None, int act=0, float alpha=1.0, float drop_p=0.0, "
          "int drop_site=0, int drop_seed=0, bool drop_first=False) -> ()", &gemm_nt);
    m.def("adapter_residual_ln_fwd(Tensor A, Tensor R1, Tensor? R2, Tensor Wd, Tensor bd, Tensor Wu, Tensor bu, Tensor gamma, Tensor beta, float eps, int act, "
          "Tensor(a!) zp, Tensor(b!) z, Tensor(c!)? v, Tensor(d!) y, Tensor(e!) stats, Tensor? res32=None, Tensor(f!)? y32=None) -> ()", &adapter_residual_ln_fwd);
    m.def("adapter_residual_ln_bwd(Tensor dy, Tensor v, Tensor stats, Tensor gamma, Tensor? dres, Tensor zp, int act, Tensor WuT, Tensor WdT, bool inner_res, "
          "Tensor(a!) dv, Tensor(b!) dzp, Tensor(c!) dh, Tensor(d!)? dgamma=None, Tensor(e!)? dbeta=None, Tensor(f!)? dbias=None, Tensor(g!)? dbd=None, "
          "float drop_p=0.0, int drop_site=0, int drop_seed=0, bool bias_total=False, Tensor? beta_y=None) -> ()", &adapter_residual_ln_bwd);
    m.def("ln_fwd(Tensor v, Tensor? add, Tensor gamma, Tensor beta, float eps, Tensor(a!) y, Tensor(b!) stats, float drop_p=0.0, int drop_site=0, "
          "int drop_seed=0) -> ()", &ln_fwd);
    m.def("score_bce_fwd(Tensor emb, Tensor prec, Tensor log_mask, Tensor(a!) pos, Tensor(b!) neg, Tensor(c!) loss_ws, int B, int L, int E, bool cpc=False) -> ()",
          &score_bce_fwd);
    m.def("score_bce_bwd(Tensor emb, Tensor prec, Tensor log_mask, Tensor pos, Tensor neg, Tensor loss_ws, float loss_scale, Tensor? loss_scale_dev, "
          "Tensor(a!) d_prec, Tensor(b!) d_emb, int B, int L, int E, bool cpc=False) -> ()", &score_bce_bwd);
    m.def("fused_adam_step(Tensor(a!) p, Tensor g, Tensor(b!) m, Tensor(c!) v, Tensor seg_end, Tensor seg_group, Tensor group_lr, int step, float beta1=0.9, "
          "float beta2=0.999, float eps=1e-8, float grad_scale=1.0) -> ()", &fused_adam_step);
    m.def("topk_rank_eval(Tensor prec, Tensor item_emb, Tensor target, Tensor hist_ptr, Tensor hist_idx, Tensor(a!) rank) -> ()", &topk_rank_eval);
    m.def("lora_bwd(Tensor x, Tensor dqa, Tensor dqb, Tensor Aa, Tensor Ab, Tensor BTa, Tensor BTb, float scale_a, float scale_b, Tensor(a!) dAa, Tensor(b!) dAb, "
          "Tensor(c!) dBa, Tensor(d!) dBb, Tensor(e!)? dbias_a, Tensor(f!)? dbias_b, Tensor(g!) ws) -> ()", &lora_bwd);
    m.def("encoder_layer_fwd(Tensor x, Tensor[] w, Tensor[] ad1, Tensor[] ad2, Tensor[] saved, Tensor(a!) x1, Tensor(b!) x_out, Tensor? key_mask, Tensor? offsets, int n_items, int S, "
          "int n_heads, bool causal, float scale, float mask_neg, float ln_eps, float p_attn=0.0, float p_hidden=0.0, int drop_site=0, int drop_seed=0, int act1=1, int act2=1, "
          "bool q8_tiled=False) -> ()", &encoder_layer_fwd);
    m.def("encoder_layer_bwd(Tensor dx_out, Tensor x1, Tensor x_out, Tensor[] w, Tensor[] wT, Tensor[] ad1, Tensor[] ad2, Tensor[] adT1, Tensor[] adT2, Tensor[] saved, Tensor[] scratch, "
          "Tensor[] grads1, Tensor[] grads2, Tensor(a!)? dx_in, Tensor? key_mask, Tensor? offsets, int n_items, int S, int n_heads, bool causal, float scale, float mask_neg, float ln_eps, "
          "float p_attn=0.0, float p_hidden=0.0, int drop_site=0, int drop_seed=0, int act1=1, int act2=1, bool q8_tiled=False) -> ()", &encoder_layer_bwd);
    m.def("sasrec_block_fwd(Tensor x, Tensor log_mask, Tensor(a!) y, Tensor[] w, int n_heads, int F, int d, int act, bool inner_res, float eps, float mask_neg, float drop_attn=0.0, "
          "float drop_hidden=0.0, int drop_site=0, int drop_seed=0, bool train=False) -> ()", &sasrec_block_fwd);
    m.def("sasrec_block_bwd(Tensor x, Tensor log_mask, Tensor dy, Tensor(a!) dx, Tensor[] w, Tensor[] grads, int n_heads, int F, int d, int act, bool inner_res, float eps, float mask_neg, "
          "float drop_attn=0.0, float drop_hidden=0.0, int drop_site=0, int drop_seed=0, bool train=False) -> ()", &sasrec_block_bwd);
    m.def("embed_ln_fwd(Tensor ids, Tensor word, Tensor pos, Tensor type0, Tensor gamma, Tensor beta, float eps, Tensor(a!) out, int S, bool roberta=False, int pad_id=0, float drop_p=0.0, "
          "int drop_site=0, int drop_seed=0, Tensor(b!)? pre_out=None, Tensor(c!)? stats_out=None, Tensor(d!)? key_mask_out=None) -> ()", &embed_ln_fwd);
    m.def("patch_embed_fwd(Tensor img, Tensor(a!) out, int patch, Tensor? keep_idx=None) -> ()", &patch_embed_fwd);
    m.def("vit_assemble(Tensor patches, Tensor cls, Tensor pos, Tensor(a!) out, int n_items, int n_keep, Tensor? keep_idx=None, int tokens_out=0) -> ()", &vit_assemble);
    m.def("abi_version() -> int", &abi_version);
}
