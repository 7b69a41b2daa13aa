// Probe (one GPU, two processes): hipIpc mapping of a buffer of another process on the SAME device, a kernel of process B
// storing into process A's memory, and stream-ordered flags: hipStreamWriteValue32 (B, into A's flag word through the IPC
// mapping) / hipStreamWaitValue32 (A, on its own flag word).  Prints OK lines; exits non-zero on any failure.
//   hipcc --offload-arch=gfx950 -O2 -o ipc_probe ipc_probe.hip && ./ipc_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <unistd.h>
#include <sys/wait.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("[%s] %s: %s\n", who, #x, hipGetErrorString(e_)); exit(3); } } while (0)
__global__ void fill(float* p, int n, float v) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = v + i; }
__global__ void sum(const float* p, int n, double* out) { double s = 0; for (int i = 0; i < n; ++i) s += p[i]; *out = s; }
int main() {
  int ab[2], ba[2];
  if (pipe(ab) || pipe(ba)) return 2;
  const int n = 1 << 16;
  pid_t pid = fork();                    // BEFORE anything touches the GPU
  const char* who = pid ? "A" : "B";
  if (pid) {                             // ---- process A: owns the buffer and the flag
    float* buf; unsigned int* flag; double* out;
    CK(hipMalloc(&buf, n * sizeof(float))); CK(hipMalloc(&flag, 256)); CK(hipMalloc(&out, 8));
    CK(hipMemset(buf, 0, n * sizeof(float))); CK(hipMemset(flag, 0, 256)); CK(hipDeviceSynchronize());
    hipIpcMemHandle_t hb, hf;
    CK(hipIpcGetMemHandle(&hb, buf)); CK(hipIpcGetMemHandle(&hf, flag));
    if (write(ab[1], &hb, sizeof(hb)) != sizeof(hb) || write(ab[1], &hf, sizeof(hf)) != sizeof(hf)) return 2;
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    int can = 0; (void)hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0);
    printf("[A] CanUseStreamWaitValue = %d\n", can);
    CK(hipStreamWaitValue32(st, flag, 7, hipStreamWaitValueGte, 0xFFFFFFFFu));      // waits for B's stream-ordered flag
    hipLaunchKernelGGL(sum, dim3(1), dim3(1), 0, st, buf, n, out);
    CK(hipStreamSynchronize(st));
    double s; CK(hipMemcpy(&s, out, 8, hipMemcpyDeviceToHost));
    const double want = 5.0 * n + (double)n * (n - 1) / 2;
    printf("[A] sum after wait = %.1f (want %.1f) %s\n", s, want, s == want ? "OK" : "MISMATCH");
    char c = 1; if (write(ab[1], &c, 1) != 1) return 2;
    int stc = 0; waitpid(pid, &stc, 0);
    return (s == want && WIFEXITED(stc) && WEXITSTATUS(stc) == 0) ? 0 : 1;
  }
  // ---- process B: maps A's memory, stores into it from a kernel, then raises the flag in stream order
  hipIpcMemHandle_t hb, hf;
  if (read(ab[0], &hb, sizeof(hb)) != sizeof(hb) || read(ab[0], &hf, sizeof(hf)) != sizeof(hf)) return 2;
  float* rbuf; unsigned int* rflag;
  CK(hipIpcOpenMemHandle((void**)&rbuf, hb, hipIpcMemLazyEnablePeerAccess));
  CK(hipIpcOpenMemHandle((void**)&rflag, hf, hipIpcMemLazyEnablePeerAccess));
  printf("[B] mapped A's buffer and flag\n");
  hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  usleep(200000);                                                             // A is parked in its wait by now
  hipLaunchKernelGGL(fill, dim3(n / 256), dim3(256), 0, st, rbuf, n, 5.0f);
  CK(hipStreamWriteValue32(st, rflag, 7, 0));
  CK(hipStreamSynchronize(st));
  printf("[B] stored and signalled\n");
  char c; if (read(ab[0], &c, 1) != 1) return 2;
  CK(hipIpcCloseMemHandle(rbuf)); CK(hipIpcCloseMemHandle(rflag));
  return 0;
}
