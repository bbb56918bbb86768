// gemm.hip -- batched dense contraction entry point (linear layers, self-attention products)
// Replaces nn.Linear (reference src/utils/ops.py:187-188,219-220) and torch.bmm (src/utils/ops.py:93,100).
#include "gemm_core.h"
#include "../../include/sgamd.h"

static inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

template <typename T, class LP, class LQ, bool TR>
static int gemm_launch(const LP& lp, const LQ& lq, const Epilogue<T>& e, const sg_gemm_desc* d, hipStream_t st) {
  const int I = d->I, J = d->J, K = d->K;
  const int splits = d->splits > 1 ? d->splits : 1;
  const int prof = sg_prof_begin(st, 2.0 * (double)I * (double)J * (double)K * (double)d->batch, 2);
  if (I <= 32) sg_launch_gemm<T, LP, LQ, 32, 256, 1, 4, TR>(lp, lq, e, I, J, K, splits, d->batch, st);
  else if (I % 128 != 0 && (I % 96 == 0 || (I < 128 && I > 64))) sg_launch_gemm<T, LP, LQ, 96, 256, 1, 4, TR>(lp, lq, e, I, J, K, splits, d->batch, st);
  else sg_launch_gemm<T, LP, LQ, 128, 128, 2, 2, TR>(lp, lq, e, I, J, K, splits, d->batch, st);
  sg_prof_end(st, prof);
  SG_LAUNCH_CHECK();
  return 0;
}

template <typename T, bool F> static void fill_kc(StridedKC<T, F>& l, const void* p, long long bs, int ld, int rows, int K) {
  l.base = (const T*)p; l.bstride = bs; l.ld = ld; l.rows = rows; l.K = K;
  l.vec_ok = (K % ET<T>::VEC == 0) && (ld % ET<T>::VEC == 0) && (bs % ET<T>::VEC == 0) && aligned16(p);
}
template <typename T, bool F> static void fill_mc(StridedMC<T, F>& l, const void* p, long long bs, int ld, int rows, int K) {
  l.base = (const T*)p; l.bstride = bs; l.ld = ld; l.rows = rows; l.K = K;
  l.vec_ok = (rows % ET<T>::VEC == 0) && (ld % ET<T>::VEC == 0) && (bs % ET<T>::VEC == 0) && aligned16(p);
}

template <typename T, bool TR, bool FAST> static int gemm_forms(const Epilogue<T>& e, const sg_gemm_desc* d, hipStream_t st) {
  typedef StridedKC<T, FAST> KCL;
  typedef StridedMC<T, FAST> MCL;
  if (d->p_form == 0 && d->q_form == 0) {
    KCL lp, lq; fill_kc<T>(lp, d->p, d->p_bstride, d->ldp, d->I, d->K); fill_kc<T>(lq, d->q, d->q_bstride, d->ldq, d->J, d->K);
    return gemm_launch<T, KCL, KCL, TR>(lp, lq, e, d, st);
  } else if (d->p_form == 0 && d->q_form == 1) {
    KCL lp; MCL lq; fill_kc<T>(lp, d->p, d->p_bstride, d->ldp, d->I, d->K); fill_mc<T>(lq, d->q, d->q_bstride, d->ldq, d->J, d->K);
    return gemm_launch<T, KCL, MCL, TR>(lp, lq, e, d, st);
  } else if (d->p_form == 1 && d->q_form == 0) {
    MCL lp; KCL lq; fill_mc<T>(lp, d->p, d->p_bstride, d->ldp, d->I, d->K); fill_kc<T>(lq, d->q, d->q_bstride, d->ldq, d->J, d->K);
    return gemm_launch<T, MCL, KCL, TR>(lp, lq, e, d, st);
  } else {
    MCL lp, lq; fill_mc<T>(lp, d->p, d->p_bstride, d->ldp, d->I, d->K); fill_mc<T>(lq, d->q, d->q_bstride, d->ldq, d->J, d->K);
    return gemm_launch<T, MCL, MCL, TR>(lp, lq, e, d, st);
  }
}

template <typename T, bool TR> static int gemm_t(const sg_gemm_desc* d, hipStream_t st) {
  Epilogue<T> e;
  e.out = d->out; e.out_bstride = d->out_bstride; e.ldo = d->ldo; e.bias = d->bias;
  e.res = d->res; e.res_bstride = d->res_bstride; e.ldr = d->ldr; e.beta = d->beta;
  e.mask = nullptr; e.mask_bstride = 0; e.ldm = 0; e.split_stride = 0;
  e.alpha = d->alpha; e.alpha_ptr = d->alpha_ptr; e.flags = d->epi_flags; e.I = d->I; e.J = d->J;
  if (d->splits > 1) SG_CHECK(d->epi_flags & SG_EPI_ATOMIC, "sg_gemm: split-K needs the atomic epilogue");
  const int V = ET<T>::VEC;
  auto vec = [&](int form, const void* p, long long bs, int ld, int rows) {
    const bool common = (ld % V == 0) && (bs % V == 0) && aligned16(p);
    return common && (form == 0 ? (d->K % V == 0) : (rows % V == 0));
  };
  if (vec(d->p_form, d->p, d->p_bstride, d->ldp, d->I) && vec(d->q_form, d->q, d->q_bstride, d->ldq, d->J))
    return gemm_forms<T, TR, true>(e, d, st);
  return gemm_forms<T, TR, false>(e, d, st);
}

extern "C" int sg_gemm(const sg_gemm_desc* d, sg_stream_t stream) {
  SG_CHECK(d && d->p && d->q && d->out, "sg_gemm: null pointer");
  SG_CHECK(d->I > 0 && d->J > 0 && d->K > 0 && d->batch > 0, "sg_gemm: bad shape");
  SG_CHECK((d->p_form | d->q_form | 1) == 1, "sg_gemm: operand form must be 0 (KC) or 1 (MC)");
  SG_CHECK(!(d->epi_flags & SG_EPI_POOL), "sg_gemm: pooling epilogue is convolution-only");
  if (d->dtype == SG_DTYPE_F32) return gemm_t<float, true>(d, (hipStream_t)stream);
  if (d->dtype == SG_DTYPE_BF16) {
    if (d->no_tr) return gemm_t<bf16_t, false>(d, (hipStream_t)stream);
    return gemm_t<bf16_t, true>(d, (hipStream_t)stream);
  }
  sg_set_error("sg_gemm: bad dtype");
  return -1;
}
