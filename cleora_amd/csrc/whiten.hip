// whiten.hip — placeholder until the MFMA Gram / projection kernels land.
#include "common.h"
namespace cleora {
uint64_t gram_workspace(uint64_t, uint32_t) { return 1; }
int launch_gram(const float *, uint64_t, uint64_t, uint32_t, const double *, double *, double *, hipStream_t) {
    set_error("centered_gram: not implemented yet");
    return CLEORA_E_INVALID;
}
int launch_project(const float *, uint64_t, uint64_t, uint32_t, const float *, const float *, uint32_t, float *, uint64_t, hipStream_t) {
    set_error("project: not implemented yet");
    return CLEORA_E_INVALID;
}
}  // namespace cleora
