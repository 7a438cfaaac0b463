// sag_doggo_lane.hip - the lane-per-env Doggo step kernel (k_step<DOGGO>: generic fused step with doggo_physics
// as its substep loop; batches above ~12k envs), in a translation unit of its own.
//
// Why: doggo_physics is a non-kernel device function with a 21.7 KB private frame, 256 VGPRs + 42 AGPRs and
// ~185 SGPRs spilled into the lanes of v252..v255, called under a partial EXEC mask.  With this toolchain
// (ROCm 7.2, AMD clang 22.0.0git) a semantically neutral source change in it (constant instead of computed
// bounding radii in two planar pair tests) produced wrong results on the GPU from the first step on, while
//   * the same source compiled for the host under ASan + UBSan + pattern-initialised locals (tests/hostemu) is
//     clean and bit-identical to the shipped kernel's GPU results,
//   * -O1, or -mllvm -amdgpu-spill-sgpr-to-vgpr=false at -O3, make the GPU results of the changed source
//     bit-identical to them as well, and the machine verifier is silent
// (tests/diag_traj.py).  That puts the fault in the backend's SGPR-spill-to-VGPR-lane path
// for this function, not in the source.  This unit is therefore built with SGPR spills going to scratch memory
// (build.py: LANE_TU_FLAGS); applied to the whole library the same switch costs the Point kernels 50 %.
#define SAG_DOGGO_LANE_TU 1
#include "sag_device.hpp"

namespace sag {

hipError_t doggo_lane_upload_model(const DgModel* m) { return hipMemcpyToSymbol(HIP_SYMBOL(g_dg), m, sizeof(DgModel)); }

void doggo_lane_launch(const StepArgs& a, int blocks, hipStream_t stream) {
  hipLaunchKernelGGL((k_step<SAG_ROBOT_DOGGO, true, true>), dim3(blocks), dim3(WAVE), 0, stream, a);
}

}  // namespace sag
