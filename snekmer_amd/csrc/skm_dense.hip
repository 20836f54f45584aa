// Dense small-basis path (placeholder until the MFMA kernel lands in this round).
#include "skm_common.h"

extern "C" int skm_count_dense(skm_ctx *, const uint8_t *, int, int, const uint8_t *, const int64_t *, int64_t, int,
                               void *, int64_t)
{
    skm_set_error("skm_count_dense: not built yet");
    return SKM_E_UNSUPPORTED;
}

extern "C" int skm_cosine_dense_i8(skm_ctx *, int64_t, int64_t, int64_t, const int8_t *, const int8_t *, const float *,
                                   const float *, int, float *, int64_t)
{
    skm_set_error("skm_cosine_dense_i8: not built yet");
    return SKM_E_UNSUPPORTED;
}
