// tg_splitk.hip -- launch path of w4_gemm_kernel (reference numerics, small launches); see tg_common.cuh
#include "tg_common.cuh"
namespace {
#include "w4_gemm.cuh"
template <typename DT, bool LAYOUT_A, int CANON, bool QMX>
int go(int waves, const GemmParams& p, dim3 grid, hipStream_t st) {
  if (waves == 16) hipLaunchKernelGGL((w4_gemm_kernel<DT, LAYOUT_A, CANON, QMX, 16, 2, 4>), grid, dim3(16 * 64), 0, st, p);
  else hipLaunchKernelGGL((w4_gemm_kernel<DT, LAYOUT_A, CANON, QMX, 8, 2, 4>), grid, dim3(8 * 64), 0, st, p);
  return launch_status();
}
template <typename DT, bool LAYOUT_A, int CANON>
int go_q(bool qmx, int waves, const GemmParams& p, dim3 grid, hipStream_t st) {
  if constexpr (!std::is_same<DT, BF16>::value) {
    if (qmx) return TG_E_DTYPE;
    return go<DT, LAYOUT_A, CANON, false>(waves, p, grid, st);
  } else {
    return qmx ? go<DT, LAYOUT_A, CANON, true>(waves, p, grid, st) : go<DT, LAYOUT_A, CANON, false>(waves, p, grid, st);
  }
}
template <typename DT, bool LAYOUT_A>
int go_c(int canon, bool qmx, int waves, const GemmParams& p, dim3 grid, hipStream_t st) {
  switch (canon) {
    case CANON_NONE: return go_q<DT, LAYOUT_A, CANON_NONE>(qmx, waves, p, grid, st);
    case CANON_PAIR: return go_q<DT, LAYOUT_A, CANON_PAIR>(qmx, waves, p, grid, st);
    default: return go_q<DT, LAYOUT_A, CANON_QUAD>(qmx, waves, p, grid, st);
  }
}
}  // namespace
int tgx::splitk(int dt, bool layout_a, int canon, bool qmx, int waves, const GemmParams& p, dim3 grid, hipStream_t st) {
  if (dt == TG_BF16) return layout_a ? go_c<BF16, true>(canon, qmx, waves, p, grid, st) : go_c<BF16, false>(canon, qmx, waves, p, grid, st);
  return layout_a ? go_c<F16, true>(canon, qmx, waves, p, grid, st) : go_c<F16, false>(canon, qmx, waves, p, grid, st);
}
