// prefill_attn.hip — varlen causal prefill attention (a9).  Placeholder entry point until the
// flash-style kernel lands; fails loudly instead of computing anything else.
#include "common.h"
extern "C" int glm_varlen_attn_prefill(const void* q, const void* k, const void* v,
                                       const int32_t* cu_seqlens_q, const int32_t* cu_seqlens_k,
                                       const int32_t* block_tables, int64_t bt_stride, void* out,
                                       int64_t B, int max_seqlen_q, int nh, int nkv, int hd,
                                       int block_size, float scale, glm_stream_t stream) {
    glm_set_error("glm_varlen_attn_prefill: not implemented yet");
    return GLM_ERR_UNSUPPORTED;
}
