// gn_hibf.hip -- HIBF device path (placeholder until the level kernels land; fails loudly).
#include "gn_internal.h"

int gn_hibf_build(gn_filter*, uint32_t, const gn_ibf_desc*, const int64_t* const*, const int64_t* const*, uint64_t)
{
    return gn_fail(GN_ERANGE, "HIBF device path not built yet");
}
int gn_hibf_classify(gn_stream*, gn_filter*, hipStream_t)
{
    return gn_fail(GN_ERANGE, "HIBF device path not built yet");
}
int gn_hibf_dense(gn_stream*, uint32_t, uint32_t, uint16_t*)
{
    return gn_fail(GN_ERANGE, "HIBF device path not built yet");
}
