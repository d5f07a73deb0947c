// csrc/h16.hip once more for bfloat16 NHWC tensors: the training step's 3x3 forward / input-gradient convolutions on the fp16
// engine's window + weight-stream kernel (otp_hb_* of csrc/hb.h, reached through csrc/nhwc.hip's otp_nhwc_conv_*).
#define OTP_H16_BF16
#include "h16.hip"
