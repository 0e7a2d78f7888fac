// gwtf_api.hip -- size queries, version and error text of the C ABI (include/gwtf.h).
#include <hip/hip_runtime.h>
#include "gwtf_layout.h"
#include "../../include/gwtf.h"

extern "C" int gwtf_abi_version(void) { return GWTF_ABI_VERSION; }

extern "C" const char* gwtf_error_string(int code) {
  if (code == 0) return "success";
  if (code == GWTF_E_BADARG) return "gwtf: bad argument (null pointer, non-positive size, width beyond what the kernels reach -- eval forward f <= 128, train / backward f <= 96 --, bad mode / pattern, or B too large for train-mode FiLM)";
  if (code == GWTF_E_UNSUPPORTED) return "gwtf: no kernel instantiation was built for this layer-width list";
  return hipGetErrorString((hipError_t)code);
}

extern "C" int gwtf_padded_width(int f) { return (f + 15) / 16 * 16; }
extern "C" size_t gwtf_raw_coupling_floats(int f, int G) { return GwtfRaw(f, G).coupling_size(); }
extern "C" size_t gwtf_packed_w_coupling_floats(int f) { return GwtfPackW(gwtf_padded_width(f)).coupling_size(); }
extern "C" size_t gwtf_packed_film_coupling_floats(int f, int G) { return GwtfPackF(gwtf_padded_width(f), G).coupling_size(); }
extern "C" size_t gwtf_film_out_floats(int f) { return gwtf_film_out_size(gwtf_padded_width(f)); }

// Diagnostic: one wall-clock stamp (100 MHz constant counter) written by a one-thread kernel on `stream` -- placed between the
// nodes of a captured hipGraph it shows where parallel branches really run (tools/diag/graph_stamps.py).
namespace {
__global__ void stamp_kernel(unsigned long long* slot) { *slot = wall_clock64(); }
}  // namespace
extern "C" int gwtf_diag_stamp(unsigned long long* slot, void* stream) {
  if (!slot) return GWTF_E_BADARG;
  hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, slot);
  return (int)hipGetLastError();
}
