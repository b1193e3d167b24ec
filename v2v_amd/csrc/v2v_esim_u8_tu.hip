// uint8-input instantiations of the fused ESIM kernel (owns the float64 log table)
#define V2V_ESIM_IN kInU8
#define V2V_ESIM_LAUNCH launch_esim_u8
#define V2V_ESIM_LUT_COPY lut_esim64_copy
#define V2V_ESIM_LUT_DEV g_lut_esim64
#define V2V_ESIM_LUT_HOST kLutEsim64
#include "v2v_esim_tu.inc"
