// vr_pathtrace.hip -- one variant of the path-tracing kernel per compilation (-DVR_PT_VARIANT=0..3), so that the variants
// build in parallel and each kernel only carries the code and registers of what its scenes use:
//   0  DDA trackers, brick density grid, no emission grid      (BASELINE configs c1, c2, c3)
//   1  DDA trackers, dense fp16 density grid, no emission grid (c4)
//   2  DDA trackers, emission grid bound                       (c5; density bricks or dense, decided at run time)
//   3  global-majorant trackers (common.glsl:333-394, the code the reference compiles out with USE_DDA); everything else at run time
// Each exports pt_variant_<n>: launch + resident-block query for {no TF, TF} x {plain, STATS}.
#include "vr_pathtrace.h"

#ifndef VR_PT_VARIANT
#error "compile with -DVR_PT_VARIANT=0..3"
#endif

namespace vr {

#if VR_PT_VARIANT == 0
template <bool TF> using Cfg = TraceCfg<TF, 0, 0, 0>;
#elif VR_PT_VARIANT == 1
template <bool TF> using Cfg = TraceCfg<TF, 0, 0, 1>;
#elif VR_PT_VARIANT == 2
template <bool TF> using Cfg = TraceCfg<TF, 0, 1, 2>;
#else
template <bool TF> using Cfg = TraceCfg<TF, 1, 2, 2>;
#endif

#define VR_PT_CAT2(a, b) a##b
#define VR_PT_CAT(a, b) VR_PT_CAT2(a, b)

typedef void (*PtKernel)(const KernelArgs);
static PtKernel pick(bool tf, bool stats) {
    return tf ? (stats ? pathtrace_kernel<Cfg<true>, true> : pathtrace_kernel<Cfg<true>, false>)
              : (stats ? pathtrace_kernel<Cfg<false>, true> : pathtrace_kernel<Cfg<false>, false>);
}

int VR_PT_CAT(pt_occupancy_variant_, VR_PT_VARIANT)(bool tf, bool stats) {
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, pick(tf, stats), 256, 0) != hipSuccess || per_cu <= 0) per_cu = 4;
    return per_cu;
}
void VR_PT_CAT(pt_launch_variant_, VR_PT_VARIANT)(bool tf, bool stats, unsigned grid, hipStream_t stream, const SceneParams& P, float* sbuf, float* cold_ws,
                                                   const LaunchDesc& D, const SchedParams& S, uint32_t* status, unsigned long long* stats_buf) {
    KernelArgs A;
    A.P = P; A.D = D; A.S = S; A.sbuf = sbuf; A.cold_ws = cold_ws; A.status = status; A.stats = stats_buf;
    hipLaunchKernelGGL(pick(tf, stats), dim3(grid), dim3(256), 0, stream, A);
}

}  // namespace vr
