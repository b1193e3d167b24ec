// float32-input instantiations of the fused ESIM kernel (owns the float32 log table)
#define V2V_ESIM_IN kInF32
#define V2V_ESIM_LAUNCH launch_esim_f32
#define V2V_ESIM_LUT_COPY lut_esim32_copy
#define V2V_ESIM_LUT_DEV g_lut_esim32
#define V2V_ESIM_LUT_HOST kLutEsim32
#include "v2v_esim_tu.inc"
