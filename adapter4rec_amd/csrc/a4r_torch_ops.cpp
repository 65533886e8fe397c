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
//   torch.ops.a4r.abi_version              a4r_version
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
    m.def("abi_version() -> int", &abi_version);
}
