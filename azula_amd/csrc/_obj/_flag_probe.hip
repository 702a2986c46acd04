#include <hip/hip_runtime.h>
__global__ void az_flag_probe() {}
