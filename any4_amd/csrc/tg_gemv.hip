// tg_gemv.hip -- launch path of w4_gemv_kernel (one layer per launch, m <= 4: the decode step's GEMMs); see tg_common.cuh
#include "tg_common.cuh"
int tgx::gemv(int dt, int I, bool qmx, GemmParams& p, int64_t batch, hipStream_t st) { return TG_PAIR_NA; }
