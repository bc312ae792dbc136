// Single translation unit of libvbmc_hip.so: the kernels live in headers shared by both ABI files.
#include "abi_elbo.hip"
#include "abi_gp.hip"
#include "abi_comm.hip"
