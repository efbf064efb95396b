// Library-level entry points: version, last-error string, device probe.
#include "ppk_common.h"
#include "ppk_kernels.h"
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

static thread_local char g_err[512] = "";

void ppk_set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" {

const char* ppk_version(void) { return "ppk 0.1.0 (gfx950)"; }

const char* ppk_last_error(void) { return g_err; }

// Number of visible HIP devices, or a negative PPK_ERR_HIP.  Lets the host
// side fail loudly when the library is loaded on a box without a GPU.
int ppk_device_count(void)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        ppk_set_error("hipGetDeviceCount: %s", hipGetErrorString(e));
        return PPK_ERR_HIP;
    }
    return n;
}

}  // extern "C"

// ---------------------------------------------------------------- profiling
namespace {
struct ProfRec { hipEvent_t a, b; double work; int kind; };
bool g_prof_on = false;
std::vector<ProfRec> g_prof;
const char* const kKindNames[ppk::PK_COUNT] = {
    "k_gemm<64,64,XX>", "k_gemm<64,64,XK>", "k_gemm<64,64,KX>", "k_gemm<64,64,KK>",
    "k_gemm<128,128,XX>", "k_gemm<128,128,XK>", "k_gemm<128,128,KX>", "k_gemm<128,128,KK>",
    "k_stack_fwd", "k_stack_bwd",
    "k_terms", "k_dz", "k_spmm", "k_graphreg_sparse",
    "k_nt64", "k_ntF", "k_nt32", "k_ntFs", "k_tn128", "k_seq_fwd", "k_seq_bwd",
};
}  // namespace

namespace ppk {
ProfScope::ProfScope(int kind, double work, hipStream_t s) : idx(-1), stream(s)
{
    if (!g_prof_on) return;
    ProfRec r;
    r.work = work;
    r.kind = kind;
    if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return;
    (void)hipEventRecord(r.a, s);
    g_prof.push_back(r);
    idx = (int)g_prof.size() - 1;
}
ProfScope::~ProfScope()
{
    if (idx >= 0) (void)hipEventRecord(g_prof[(size_t)idx].b, stream);
}
}  // namespace ppk

extern "C" {

int ppk_prof_enable(int on)
{
    g_prof_on = on != 0;
    return PPK_OK;
}

int ppk_prof_kind_count(void) { return ppk::PK_COUNT; }

const char* ppk_prof_kind_name(int kind) { return (kind >= 0 && kind < ppk::PK_COUNT) ? kKindNames[kind] : ""; }

// Waits for the recorded events, sums per kind (ms, work, launches; arrays of
// ppk_prof_kind_count() entries), then clears the records.
int ppk_prof_collect(double* ms, double* work, int64_t* launches)
{
    if (!ms || !work || !launches) PPK_FAIL(PPK_ERR_NULL, "prof_collect: null pointer");
    for (int k = 0; k < ppk::PK_COUNT; ++k) { ms[k] = 0; work[k] = 0; launches[k] = 0; }
    const char* dump = getenv("PPK_PROF_DUMP");          // tuning aid: one "kind work ms" line per launch
    FILE* df = dump ? fopen(dump, "a") : nullptr;
    for (ProfRec& r : g_prof) {
        float t = 0.f;
        if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&t, r.a, r.b) == hipSuccess) {
            ms[r.kind] += t;
            work[r.kind] += r.work;
            launches[r.kind] += 1;
            if (df) fprintf(df, "%s %.0f %.5f\n", kKindNames[r.kind], r.work, (double)t);
        }
        (void)hipEventDestroy(r.a);
        (void)hipEventDestroy(r.b);
    }
    g_prof.clear();
    if (df) fclose(df);
    return PPK_OK;
}

}  // extern "C"
