// Host-side probe of the split-bf16 helpers the kernels and the host share (harmony_amd/csrc/hmx_internal.h): built by
// tests/test_abi_cpu.py with hipcc (host code only, no GPU needed) and called through ctypes.
#include "../../harmony_amd/csrc/hmx_internal.h"
extern "C" {
void probe_bf3_split(float x, unsigned short* parts) { unsigned short p[3]; hmx::bf3_split(x, p); parts[0] = p[0]; parts[1] = p[1]; parts[2] = p[2]; }
long long probe_bfimg_index(int nct, int ns2, int j, int k, int part) { return (long long)hmx::bfimg_index(nct, ns2, j, k, part); }
int probe_kcol(int nct, int ct, int c) { return hmx::kcol(nct, ct, c); }
}
