// Sanitizer build of the library's HOST code only (make asan; tools/asan_host.sh): the kernel launchers of fx_kernels.hip are
// replaced by stubs that report hipErrorNotSupported -- nothing on a CPU box launches a kernel; what runs under ASan / UBSan is
// the host logic of fx_api.hip that needs no GPU (fx_pack_predictions, fx_invert_cov2, fx_build_obstacle_hulls*, fx_cs_to_curvilinear*,
// fx_build_boundary_bins, fx_wait_word, argument validation) and the CPython extension _fxhost (fx_host_ext.c).
#include <hip/hip_runtime_api.h>
#define FX_STUB(name) extern "C" hipError_t name(...) { return hipErrorNotSupported; }
FX_STUB(fx_launch_eval) FX_STUB(fx_launch_eval_grid) FX_STUB(fx_launch_obstacle) FX_STUB(fx_step_kernel_capacity) FX_STUB(fx_launch_step)
FX_STUB(fx_launch_select) FX_STUB(fx_launch_math_test) FX_STUB(fx_launch_publish) FX_STUB(fx_launch_stage) FX_STUB(fx_launch_package)
FX_STUB(fx_launch_topk) FX_STUB(fx_launch_probe_read)
