// csrc/densex.hip once more with bfloat16 operand pieces: otp_dense_x3_pack_bf16p / otp_dense_x3_bf16p, the C -> C projection
// for operands of unknown magnitude (the training backward's dx = W^T dy) - see the head of densex.hip.
#define OTP_X3_BF16
#define OTP_X3_GRAD_COPY
#define OTP_ENTRY(name) name##_bf16p
#include "densex.hip"
