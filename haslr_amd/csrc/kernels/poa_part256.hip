// K6 instances with workgroups of up to 256 lanes (kernels/poa.hip is compiled in four parts: see poa_run there)
#define HX_POA_PART 256
#include "poa.hip"
