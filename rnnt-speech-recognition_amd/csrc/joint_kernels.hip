// joint_kernels.hip -- fused joint network + transducer loss (placeholder until the MFMA kernels land).
#include "rnnt_common.h"

namespace rnnt {

hipError_t joint_workspace_bytes(int, int, int, int, int, size_t *bytes) {
    *bytes = 0;
    return hipErrorNotSupported;
}

hipError_t launch_joint_loss(const float *, const float *, const float *, const float *, const int *, const int *,
                             const int *, const float *, int, int, int, int, int, int, float *, float *, float *,
                             float *, float *, int, void *, hipStream_t) {
    return hipErrorNotSupported;
}

}  // namespace rnnt
