// Link-side question behind DESIGN.md "Next" #3 (2-D chunked copies so that an NTT pass can start on column blocks while the rest of the vector is still on the
// link): what does a 2-D copy of a column block cost against the linear copy of the same bytes, from / to PAGEABLE memory (what the reference hands over)
// and from / to pinned memory?  32 MiB vector seen as 1024 rows x 32 KiB; column blocks of 1 / 4 / 8 / 32 KiB per row.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/memcpy2d tools/ubench/memcpy2d.hip && /tmp/memcpy2d
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    const size_t rows = 1024, pitch = 32768, bytes = rows * pitch;
    char *d, *pin, *pg;
    CK(hipMalloc(&d, bytes));
    CK(hipHostMalloc(&pin, bytes, hipHostMallocDefault));
    pg = (char*)aligned_alloc(4096, bytes);
    memset(pg, 1, bytes); memset(pin, 2, bytes);
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    for (int dir = 0; dir < 2; dir++)
        for (int pinned = 0; pinned < 2; pinned++) {
            char* h = pinned ? pin : pg;
            for (size_t width : { (size_t)0, (size_t)1024, (size_t)4096, (size_t)8192, (size_t)32768 }) {
                double best = 1e9, sum = 0;
                const int reps = 7;
                for (int r = 0; r < reps; r++) {
                    CK(hipStreamSynchronize(st));
                    const double t0 = now();
                    if (width == 0) { // linear, whole vector
                        if (dir == 0) CK(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, st)); else CK(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, st));
                    } else {
                        for (size_t c = 0; c < pitch; c += width) { // all column blocks: the same 32 MiB
                            if (dir == 0) CK(hipMemcpy2DAsync(d + c, pitch, h + c, pitch, width, rows, hipMemcpyHostToDevice, st));
                            else CK(hipMemcpy2DAsync(h + c, pitch, d + c, pitch, width, rows, hipMemcpyDeviceToHost, st));
                        }
                    }
                    CK(hipStreamSynchronize(st));
                    const double t = now() - t0;
                    if (r > 0) { best = t < best ? t : best; sum += t; }
                }
                printf("%s %-8s %-22s 32 MiB: best %.3f ms  mean %.3f ms  (%.1f GB/s)\n", dir ? "D2H" : "H2D", pinned ? "pinned" : "pageable",
                       width == 0 ? "linear" : (width == 1024 ? "2-D blocks of 1 KiB" : width == 4096 ? "2-D blocks of 4 KiB" : width == 8192 ? "2-D blocks of 8 KiB" : "2-D blocks of 32 KiB"),
                       best, sum / (reps - 1), bytes / best / 1e6);
                fflush(stdout);
            }
        }
    return 0;
}
