// float32-input feature-specialised instances of the v2e kernel
#define V2V_V2E_SPEC_IN kInF32
#define V2V_V2E_SPEC_LAUNCH launch_v2e_spec_f32
#include "v2v_v2e_spec_tu.inc"
