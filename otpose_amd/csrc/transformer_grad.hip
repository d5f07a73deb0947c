// The two attention products of csrc/transformer.hip once more with bfloat16 operand pieces: otp_chan_attn_scores_bf16p /
// otp_chan_attn_apply_bf16p, for the training backward (dS = dO v^T, dq / dk / dv) - see the head of transformer.hip.
#define OTP_X3_BF16
#define OTP_X3_GRAD_COPY
#define OTP_ENTRY(name) name##_bf16p
#include "transformer.hip"
