// libfdcap_hip.so -- kernels + C-ABI (include/fdcap.h) of the MI355X global-optimisation path.
// gfx950 only.  See DESIGN.md for the data layout and the per-kernel rooflines.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <string>
#include <vector>

// Build requirement (the finding: NOTES.md section 6): this file is compiled WITHOUT packed fp32 instructions
//   -Xclang -target-feature -Xclang -packed-fp32-ops -DFDC_BUILD_NO_PK_F32
// (on the MI355X boxes this was developed on, v_pk_{fma,mul,add}_f32 now and then return a wrong low element while another
// kernel's MFMAs share the CU: tools/pk_f32_mfma_repro.hip).  The define travels with the flag so that a build script which
// drops one drops both and stops here; capi.load_library() reads it back through fdcap_build_info(), and
// tests/test_io_and_abi.py disassembles the library to check the flag really took effect.
#if !defined(FDC_BUILD_NO_PK_F32) && !defined(FDC_BUILD_ALLOW_PK_F32)
#error "build fdcap.hip with: -Xclang -target-feature -Xclang -packed-fp32-ops -DFDC_BUILD_NO_PK_F32 (see __graft_entry__.build)"
#endif

#ifdef FDC_PN_TIMING
// instrumentation build only: per-frame s_memtime stamps of the pose kernels [which][block][8]
__device__ unsigned long long g_fr_times[3][2048 * 8];
#define FDC_FR_STAMP(w, i) do { const unsigned fb_ = blockIdx.x * gridDim.y + blockIdx.y; if (threadIdx.x == 0 && fb_ < 2048) g_fr_times[w][fb_ * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#endif
#ifndef FDC_FR_STAMP
#define FDC_FR_STAMP(w, i)
#endif
#include "../../include/fdcap.h"
#include "fdc_chamfer.h"
#include "fdc_comm.h"
#include "fdc_dct.h"
#include "fdc_fit2d.h"
#include "fdc_frame.h"
#include "fdc_gemm.h"
#include "fdc_host_setup.h"
#include "fdc_lbfgs.h"
#include "fdc_loss.h"
#include "fdc_math.h"
#include "fdc_panel.h"
#include "fdc_scene.h"
#include "fdc_skin.h"
#include "fdc_trace.h"

using namespace fdc;

#define HIP_TRY(expr)                              \
    do {                                           \
        hipError_t _e = (expr);                    \
        if (_e != hipSuccess) return (int)_e;      \
    } while (0)

// The translation unit, in dependency order (each part opens and closes its own `namespace { }` / `extern "C" { }`):
#include "fdc_k_pose.h"      // per-frame pose kernels (forward, backward + parameter-space losses)
#include "fdc_k_skin.h"      // skinning forward / backward (+ contact robustifier)
#include "fdc_k_step.h"      // mode 'local' kernels, loss reductions, Adam / exchange step, conversions, operator helpers
#include "fdc_state.h"       // DevBuf, fdcap_ctx, OptState, shared host helpers
#include "fdc_api_ops.h"     // C-ABI: context, scene, contact ids, Op 1-3 (+ backward), conversions
#include "fdc_api_opt.h"     // C-ABI: the clip optimiser's iteration
#include "fdc_api_modes.h"   // C-ABI: mode 'dct', inner fit / L-BFGS, checkpoint state, per-frame smoother
#include "fdc_api_run.h"     // C-ABI: mode 'local', step / results, RCCL exchange, fdcap_opt_run, timing entry points
