// msm_g2.hip -- G2 (Fq2) instantiation of the MSM kernels (B2 query).
#include "msm_curve.inc.h"
namespace g16 {
template struct MsmPoints<Fq2>;
template struct MsmWork<Fq2>;
template void msm_run<Fq2>(const MsmSort&, const MsmPoints<Fq2>&, uint32_t, MsmWork<Fq2>&,
                           MsmAcc<Fq2>*, hipStream_t, StageTimer*);
template void msm_accumulate<Fq2>(const MsmSort&, const MsmPoints<Fq2>&, uint32_t, MsmWork<Fq2>&, int,
                                 hipStream_t, StageTimer*, bool);
template void msm_fixup<Fq2>(const MsmSort&, const MsmPoints<Fq2>&, uint32_t, MsmWork<Fq2>&, int, hipStream_t, StageTimer*);
template void msm_reduce<Fq2>(const MsmSort&, MsmWork<Fq2>&, int, int, MsmAcc<Fq2>*, hipStream_t,
                             StageTimer*, bool);
}  // namespace g16
