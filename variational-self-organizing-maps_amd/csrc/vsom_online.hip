// vsom_online.hip -- online path (Som::trainSingle, Som.cpp:885-947).  Placeholder until the
// kernels land: the entry points fail loudly instead of falling back to a CPU path.
#include "vsom_internal.hpp"

extern "C" {
int vsom_train_single(vsom_ctx *, const float *, double, double, uint64_t *, int, float *, float *,
                      uint64_t *)
{
    return vsom_fail(VSOM_ERR_UNSUPPORTED, "vsom_train_single: not implemented yet");
}
int vsom_train_online_chunk(vsom_ctx *, double, double, int, float *)
{
    return vsom_fail(VSOM_ERR_UNSUPPORTED, "vsom_train_online_chunk: not implemented yet");
}
}
