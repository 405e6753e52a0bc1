// volren_amd.hpp -- the C++ drop-in surface: the reference's in-process object API (src/renderer.h:16-63,
// src/environment.h:7-23, src/transferfunc.h:9-42 and the slice of voldata it consumes) implemented on HIP.
//
// A caller of the reference (src/main.cpp, src/bindings.cpp) includes "renderer.h" and drives a RendererOpenGL; against this
// library it includes <volren_amd.hpp> and the same code compiles: same class names, same public fields, same call protocol
// (mutate fields -> commit() after changing the volume -> reset() -> trace() once per sample; result = running mean in
// `color`, RGBA32F, row 0 at the bottom).  Additions only: render(n) (all samples in one fused launch), an explicit camera
// and resolution (the reference reads cppgl globals), set_tiles() / ShardedRenderer for multi-GPU sharding, fast_math, integrator.
// Build: hipcc (the headers include <hip/hip_runtime.h> for the device-buffer handles); link libvolren_amd.so.
// FFI users bind the C ABI in volren_amd.h instead; INTEGRATION.md shows both.
//
// Two layouts, one file.  Installed (`make install PREFIX=<p>`): <p>/include/volren_amd.h, <p>/include/volren_amd.hpp and the class headers under
// <p>/include/volren_amd/, <p>/lib/libvolren_amd.so, <p>/bin/volren -- a caller builds with `hipcc -I<p>/include ... -L<p>/lib -lvolren_amd`.
// Source tree: this file sits in include/ and the class headers in volren_amd/csrc/.
#pragma once

#if __has_include("volren_amd/renderer.h")
#include "volren_amd/renderer.h"                // RendererHIP, Camera, BrickGridHIP
#include "volren_amd/environment.h"             // Environment
#include "volren_amd/transferfunc.h"            // TransferFunction
#include "volren_amd/grids.h"                   // Volume, Grid, DenseGrid, DenseGridF16, BrickGrid, Buf3D
#include "volren_amd/sharded.h"                 // ShardedRenderer: one frame on several devices (no reference counterpart)
#else
#include "../volren_amd/csrc/renderer.h"
#include "../volren_amd/csrc/environment.h"
#include "../volren_amd/csrc/transferfunc.h"
#include "../volren_amd/csrc/grids.h"
#include "../volren_amd/csrc/sharded.h"
#endif

// the reference's names
using RendererOpenGL = vr::RendererHIP;         // src/renderer.h:16
using Environment = vr::Environment;            // src/environment.h:7
using TransferFunction = vr::TransferFunction;  // src/transferfunc.h:9
using ShardedRenderer = vr::ShardedRenderer;    // addition: N RendererOpenGL parts on N devices, one gather per frame
namespace voldata {
using Volume = vr::Volume;
using Grid = vr::Grid;
using DenseGrid = vr::DenseGrid;
using BrickGrid = vr::BrickGrid;
template <typename T> using Buf3D = vr::Buf3D<T>;
}  // namespace voldata
