// K6 instances with workgroups of up to 64 lanes (kernels/poa.hip is compiled in four parts: see poa_run there)
#define HX_POA_PART 64
#include "poa.hip"
