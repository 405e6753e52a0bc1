// vr_pathtrace.hip -- one variant of the path-tracing kernel per compilation (-DVR_PT_VARIANT=0..3), so that the variants
// build in parallel and each kernel only carries the code and registers of what its scenes use:
//   0  DDA trackers, brick density grid, no emission grid      (BASELINE configs c1, c2, c3)
//   1  DDA trackers, dense fp16 density grid, no emission grid (c4)
//   2  DDA trackers, brick density grid, emission grid bound   (c5)
//   4  variant 2 for grids whose majorant table keeps levels 0-1 in 4x4x4-cell blocks (GridView::maj_blocked, chosen per grid at commit(): the large, well
//      filled sparse grids of BASELINE configs[4], whose DDA walks then touch a quarter fewer cache lines)
//   3  everything decided at run time: the global-majorant trackers (common.glsl:333-394, the code the reference compiles out with
//      USE_DDA) and the one remaining combination, a dense fp16 density grid with an emission grid
// Each is built twice: bit-exact arithmetic (the default and the parity target) and, with -DVR_FAST_MATH=1, the opt-in
// tolerance mode (hardware transcendentals, reciprocal division, contraction; vr_math.h).  A compilation exports two C symbols,
// vr_pt_occupancy_<variant>[_fast] and vr_pt_launch_<variant>[_fast], for {no TF, TF} x {plain, STATS}.
#include <hip/hip_runtime.h>
#include <stdint.h>

#ifndef VR_PT_VARIANT
#error "compile with -DVR_PT_VARIANT=0..4"
#endif
#ifdef VR_FAST_MATH
#define vr vr_fastmath            // the inline lane code differs from the exact build's: keep the two apart for the linker
#define VR_PT_SUFFIX _fast
#else
#define VR_PT_SUFFIX
#endif

#if VR_PT_VARIANT == 3 && !defined(VR_BATCH_REGS)
// the everything-at-run-time variant needs every register it can get: its event batches reuse the marching path's registers (the
// marching path waits in its LDS slot meanwhile) instead of a second set -- no scratch, at the price of an LDS round trip per batch
#define VR_BATCH_REGS 0
#endif
#if VR_PT_VARIANT == 3 && !defined(VR_HOT_PAIRS)
// ... and one copy of the hot pair per scheduler iteration (the other variants run up to four, vr_pathtrace.h): a second copy costs its transfer-function
// instance 8 VGPR spills to scratch memory (profiles/r5_kernel_resources.txt)
#define VR_HOT_PAIRS 1
#endif
#if VR_PT_VARIANT == 3 && !defined(VR_CLEAN_FORMS)
// ... and the hot pair in its general form only (vr_trace.h seg_clean: the other variants also carry the form for wavefronts whose paths are all on clean segments)
#define VR_CLEAN_FORMS 0
#endif
#include "vr_pathtrace.h"

namespace vr {

#if VR_PT_VARIANT == 0
template <bool TF> using Cfg = TraceCfg<TF, 0, 0, 0>;
#elif VR_PT_VARIANT == 1
template <bool TF> using Cfg = TraceCfg<TF, 0, 0, 1>;
#elif VR_PT_VARIANT == 2
template <bool TF> using Cfg = TraceCfg<TF, 0, 1, 0>;
#elif VR_PT_VARIANT == 4
template <bool TF> using Cfg = TraceCfg<TF, 0, 1, 0, 1>;
#else
template <bool TF> using Cfg = TraceCfg<TF, 2, 2, 2, 2>;
#endif

typedef void (*PtKernel)(const KernelArgs);
static PtKernel pick(bool tf, bool stats) {
    return tf ? (stats ? pathtrace_kernel<Cfg<true>, true> : pathtrace_kernel<Cfg<true>, false>)
              : (stats ? pathtrace_kernel<Cfg<false>, true> : pathtrace_kernel<Cfg<false>, false>);
}

}  // namespace vr

#define VR_PT_CAT3(a, b, c) a##b##c
#define VR_PT_CAT(a, b, c) VR_PT_CAT3(a, b, c)

extern "C" int VR_PT_CAT(vr_pt_occupancy_, VR_PT_VARIANT, VR_PT_SUFFIX)(int tf, int stats) {
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, vr::pick(tf != 0, stats != 0), 64 * vr::kWgWaves, 0) != hipSuccess || per_cu <= 0) per_cu = 16 / vr::kWgWaves;
    return per_cu;
}
// P, D, S: SceneParams, LaunchDesc, SchedParams (plain data, same layout in both builds)
extern "C" void VR_PT_CAT(vr_pt_launch_, VR_PT_VARIANT, VR_PT_SUFFIX)(int tf, int stats, unsigned grid, hipStream_t stream, const void* P, float* sbuf, float* cold_ws,
                                                                       const void* D, const void* S, uint32_t* status, unsigned long long* stats_buf) {
    vr::KernelArgs A;
    A.P = *static_cast<const vr::SceneParams*>(P);
    A.D = *static_cast<const vr::LaunchDesc*>(D);
    A.S = *static_cast<const vr::SchedParams*>(S);
    A.sbuf = sbuf; A.cold_ws = cold_ws; A.status = status; A.stats = stats_buf;
    hipLaunchKernelGGL(vr::pick(tf != 0, stats != 0), dim3(grid), dim3(64 * vr::kWgWaves), 0, stream, A);
}
