// placeholder until the Leiden kernels land (first GPU bring-up run only)
#include "common.h"
using namespace scamd;
extern "C" size_t scamd_leiden_workspace_bytes(int64_t n, int64_t nnz) { return 0; }
extern "C" int scamd_leiden_csr_f32(const int64_t*, const int32_t*, const float*, int64_t, int64_t, double, int,
                                    double, uint64_t, int32_t*, double*, int32_t*, void*, size_t, scamd_stream_t) {
  set_error("leiden: not built yet");
  return SCAMD_EUNSUPPORTED;
}
extern "C" int scamd_modularity_csr_f32(const int64_t*, const int32_t*, const float*, int64_t, const int32_t*, double,
                                        double*, void*, size_t, scamd_stream_t) {
  set_error("modularity: not built yet");
  return SCAMD_EUNSUPPORTED;
}
