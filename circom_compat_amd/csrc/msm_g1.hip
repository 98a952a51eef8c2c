// msm_g1.hip -- G1 (Fq) instantiation of the MSM kernels (A, B1, L, H queries).
#include "msm_curve.inc.h"
namespace g16 {
template struct MsmPoints<Fq>;
template struct MsmWork<Fq>;
template void msm_run<Fq>(const MsmSort&, const MsmPoints<Fq>&, uint32_t, MsmWork<Fq>&, MsmAcc<Fq>*,
                          hipStream_t, StageTimer*);
template void msm_accumulate<Fq>(const MsmSort&, const MsmPoints<Fq>&, uint32_t, MsmWork<Fq>&, int,
                                 hipStream_t, StageTimer*, bool);
template void msm_fixup<Fq>(const MsmSort&, const MsmPoints<Fq>&, uint32_t, MsmWork<Fq>&, int, hipStream_t, StageTimer*);
template void msm_accumulate_pair<Fq>(const MsmSort&, const MsmPoints<Fq>&, const MsmPoints<Fq>&,
                                      MsmWork<Fq>&, int, hipStream_t, StageTimer*, bool);
template void msm_fixup_pair<Fq>(const MsmSort&, const MsmPoints<Fq>&, const MsmPoints<Fq>&, MsmWork<Fq>&, int,
                                 hipStream_t, StageTimer*);
template void msm_reduce<Fq>(const MsmSort&, MsmWork<Fq>&, int, int, MsmAcc<Fq>*, hipStream_t,
                             StageTimer*, bool);
}  // namespace g16
