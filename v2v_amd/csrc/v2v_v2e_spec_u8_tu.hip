// uint8-input feature-specialised instances of the v2e kernel
#define V2V_V2E_SPEC_IN kInU8
#define V2V_V2E_SPEC_LAUNCH launch_v2e_spec_u8
#include "v2v_v2e_spec_tu.inc"
