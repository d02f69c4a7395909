// decode_kernels.hip -- explicit instantiations of the decode kernel (decode_kernel.h), one group per translation unit so
// that the groups compile in parallel: hipcc ... -DCTC_KERNEL_GROUP=<g> for g in [0, CTC_KERNEL_GROUPS).
#include "decode_kernel.h"

#ifndef CTC_KERNEL_GROUP
#error "compile with -DCTC_KERNEL_GROUP=<g>"
#endif

namespace ctcdk {

template <int G> struct InGroup { static constexpr bool value = G == (CTC_KERNEL_GROUP); };

#define CTC_X_INST(PROF_, BIG_, LAYOUT_, PRUNED_, NT_, LM_, OCC2_, G_) CTC_INST_##G_(PROF_, BIG_, LAYOUT_, PRUNED_, NT_, LM_, OCC2_)
#define CTC_DO_INST(PROF_, BIG_, LAYOUT_, PRUNED_, NT_, LM_, OCC2_) template __global__ void ctc_beam_decode_kernel<PROF_, BIG_, LAYOUT_, PRUNED_, NT_, LM_, OCC2_>(KernelArgs);
#if CTC_KERNEL_GROUP == 0
#define CTC_INST_0(...) CTC_DO_INST(__VA_ARGS__)
#else
#define CTC_INST_0(...)
#endif
#if CTC_KERNEL_GROUP == 1
#define CTC_INST_1(...) CTC_DO_INST(__VA_ARGS__)
#else
#define CTC_INST_1(...)
#endif
#if CTC_KERNEL_GROUP == 2
#define CTC_INST_2(...) CTC_DO_INST(__VA_ARGS__)
#else
#define CTC_INST_2(...)
#endif
#if CTC_KERNEL_GROUP == 3
#define CTC_INST_3(...) CTC_DO_INST(__VA_ARGS__)
#else
#define CTC_INST_3(...)
#endif
#if CTC_KERNEL_GROUP == 4
#define CTC_INST_4(...) CTC_DO_INST(__VA_ARGS__)
#else
#define CTC_INST_4(...)
#endif
#if CTC_KERNEL_GROUP == 5
#define CTC_INST_5(...) CTC_DO_INST(__VA_ARGS__)
#else
#define CTC_INST_5(...)
#endif
#if CTC_KERNEL_GROUP == 6
#define CTC_INST_6(...) CTC_DO_INST(__VA_ARGS__)
#else
#define CTC_INST_6(...)
#endif
#if CTC_KERNEL_GROUP == 7
#define CTC_INST_7(...) CTC_DO_INST(__VA_ARGS__)
#else
#define CTC_INST_7(...)
#endif
#if CTC_KERNEL_GROUP == 8
#define CTC_INST_8(...) CTC_DO_INST(__VA_ARGS__)
#else
#define CTC_INST_8(...)
#endif
#if CTC_KERNEL_GROUP == 9
#define CTC_INST_9(...) CTC_DO_INST(__VA_ARGS__)
#else
#define CTC_INST_9(...)
#endif
#if CTC_KERNEL_GROUP == 10
#define CTC_INST_10(...) CTC_DO_INST(__VA_ARGS__)
#else
#define CTC_INST_10(...)
#endif
#if CTC_KERNEL_GROUP == 11
#define CTC_INST_11(...) CTC_DO_INST(__VA_ARGS__)
#else
#define CTC_INST_11(...)
#endif

CTC_KERNEL_LIST(CTC_X_INST)

}  // namespace ctcdk
