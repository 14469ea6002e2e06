// compute_normals (pasture-algorithms/src/normal_estimation.rs:79-130) — device implementation lives in normals.hip.
#include "runtime.hpp"

using namespace pst;

extern "C" {
int pst_compute_normals(const pst_buffer*, size_t, double*, double*, int64_t*) {
  PST_API_BEGIN
  throw Error(PST_ERR_UNSUPPORTED, "pst_compute_normals: kNN normal estimation kernel not built yet");
  PST_API_END
}
int pst_compute_normals_into(const pst_buffer*, size_t, pst_buffer*) {
  PST_API_BEGIN
  throw Error(PST_ERR_UNSUPPORTED, "pst_compute_normals_into: kNN normal estimation kernel not built yet");
  PST_API_END
}
}
