// qcat_hip.hip -- libqcat_hip.so: C ABI (include/qcat_hip.h) + host orchestration.
// gfx950 only; built by __graft_entry__.build():
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared qcat_hip.hip -o libqcat_hip.so
//
// Per batch the library enqueues on the context's stream:
//   k_pack_windows -> [packed path: k_adapter_packed -> k_job_* -> k_barcode_packed |
//                      generic path: k_scan_generic] -> k_finalize
// (see DESIGN.md for the data layout and the roofline of each kernel).
#include <hip/hip_runtime.h>

#include <sched.h>

#include <algorithm>
#include <atomic>
#include <memory>
#include <cctype>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "kit.h"
#include "synth.h"

#include "kit_prepare.inc"
#include "kernels_common.inc"
#include "kernels_generic.inc"
#include "kernels_tiny.inc"
#include "kernels_packed.inc"
#include "kernels_bitslice.inc"
#include "packed_host.inc"
#include "kernels_static.inc"
#include "kernels_middle.inc"
#include "static_registry.inc"
#include "host_pipeline.inc"

using namespace qk;

// ------------------------------------------------------------------------------------------
// error handling
// ------------------------------------------------------------------------------------------
static thread_local std::string g_err;

static int set_err(int rc, const std::string& m) { g_err = m; return rc; }

#define HIPCHK(expr)                                                                         \
    do {                                                                                     \
        hipError_t e__ = (expr);                                                             \
        if (e__ != hipSuccess)                                                               \
            return set_err(QCAT_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e__)); \
    } while (0)
// ... for code with copies in flight to or from memory that dies with the function: the stream is drained before the return
#define HIPCHK_DRAIN(stream, expr)                                                           \
    do {                                                                                     \
        hipError_t e__ = (expr);                                                             \
        if (e__ != hipSuccess) {                                                             \
            (void)hipStreamSynchronize(stream);                                              \
            return set_err(QCAT_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e__)); \
        }                                                                                    \
    } while (0)

extern "C" const char* qcat_last_error(void) { return g_err.c_str(); }
extern "C" int qcat_abi_version(void) { return QCAT_ABI_VERSION; }
extern "C" const char* qcat_backend(void) { return "hip"; }

// ------------------------------------------------------------------------------------------
// options (options.h): the table, its one import from the environment, the ABI around it
// ------------------------------------------------------------------------------------------
std::atomic<int64_t> g_qcat_opt[QO_COUNT];
static const char* const g_qcat_opt_name[QO_COUNT] = {
#define X(N, D) #N,
    QCAT_OPTION_LIST(X)
    QCAT_AB_OPTION_LIST(X)
#undef X
};
static const char* const g_qcat_opt_doc[QO_COUNT] = {
#define X(N, D) D,
    QCAT_OPTION_LIST(X)
    QCAT_AB_OPTION_LIST(X)
#undef X
};
static void qcat_options_from_env() {
    for (int i = 0; i < QO_COUNT; ++i) {
        const std::string env = std::string("QCAT_HIP_") + g_qcat_opt_name[i];
        const char* e = i < QO_SETTABLE ? getenv(env.c_str()) : nullptr;     // (the library's only look at QCAT_HIP_* switches: once, here;
                                                                             //  the A/B switches of dropped variants: -DQCAT_AB builds only)
        g_qcat_opt[i].store(e ? (*e ? (int64_t)atoll(e) : 1) : QOPT_UNSET, std::memory_order_relaxed);
    }
}
// (Round 6, measured and not kept: GPU_MAX_HW_QUEUES = 12 set from here at load -- the runtime's default of four hardware queues
// lets four of a small batch's side-by-side launches run at a time.  With twelve every cross-stream dependency of a call
// becomes a cross-QUEUE signal of 40-50 us where it was an in-queue barrier of 6: the 4000-read kit-auto call 0.81 -> 1.19 ms,
// config 2 0.99 -> 1.66 ms, the dual kit 3.84 -> 5.08 ms, config 3 31.9 -> 32.8 ms; profiles/r06_ab_hw_queues.txt.)
static struct QcatOptInit { QcatOptInit() { qcat_options_from_env(); } } g_qcat_opt_init;
static int qcat_opt_index(const char* name) {
    if (!name) return -1;
    if (strncmp(name, "QCAT_HIP_", 9) == 0) name += 9;
    for (int i = 0; i < QO_SETTABLE; ++i) if (strcmp(name, g_qcat_opt_name[i]) == 0) return i;
    return -1;
}
extern "C" int qcat_option_count(void) { return QO_SETTABLE; }
extern "C" const char* qcat_option_name(int i) { return i >= 0 && i < QO_SETTABLE ? g_qcat_opt_name[i] : nullptr; }
extern "C" const char* qcat_option_doc(int i) { return i >= 0 && i < QO_SETTABLE ? g_qcat_opt_doc[i] : nullptr; }
extern "C" void qcat_reset_options(void) { qcat_options_from_env(); qk::g_alloc_gen.fetch_add(1); }
extern "C" int qcat_set_option(const char* name, int64_t value) {
    const int i = qcat_opt_index(name);
    if (i < 0) return set_err(QCAT_ERR_ARG, std::string("qcat_set_option: no option named ") + (name ? name : "(null)"));
    if (value == QOPT_UNSET) return set_err(QCAT_ERR_ARG, "qcat_set_option: the value INT64_MIN means unset (qcat_clear_option)");
    g_qcat_opt[i].store(value, std::memory_order_relaxed);
    qk::g_alloc_gen.fetch_add(1);                   // (a captured graph holds the launches the old switches chose: dropped, like after a reallocation)
    return 0;
}
extern "C" int qcat_clear_option(const char* name) {
    const int i = qcat_opt_index(name);
    if (i < 0) return set_err(QCAT_ERR_ARG, std::string("qcat_clear_option: no option named ") + (name ? name : "(null)"));
    g_qcat_opt[i].store(QOPT_UNSET, std::memory_order_relaxed);
    qk::g_alloc_gen.fetch_add(1);
    return 0;
}
extern "C" int qcat_get_option(const char* name, int64_t* value) {          // returns 1 when the option is set (*value), 0 when it is not
    const int i = qcat_opt_index(name);
    if (i < 0) return set_err(QCAT_ERR_ARG, std::string("qcat_get_option: no option named ") + (name ? name : "(null)"));
    const int64_t v = g_qcat_opt[i].load(std::memory_order_relaxed);
    if (value) *value = v == QOPT_UNSET ? 0 : v;
    return v == QOPT_UNSET ? 0 : 1;
}
extern "C" int qcat_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// NUMA node of a device's PCI function (sysfs), -1 when unknown: the rank launcher (qcat_amd/parallel.py) keeps a
// rank's host threads -- the compaction pool of host_pipeline.inc -- on the node its GPU hangs off
extern "C" int qcat_device_numa_node(int device) {
    char bdf[64] = {0};
    if (device < 0 || device >= qcat_device_count()) return -1;
    if (hipDeviceGetPCIBusId(bdf, (int)sizeof bdf, device) != hipSuccess) return -1;
    for (char* q = bdf; *q; ++q) *q = (char)tolower((unsigned char)*q);
    const std::string path = std::string("/sys/bus/pci/devices/") + bdf + "/numa_node";
    FILE* fh = fopen(path.c_str(), "r");
    if (!fh) return -1;
    int node = -1;
    if (fscanf(fh, "%d", &node) != 1) node = -1;
    fclose(fh);
    return node;
}

// host threads a context may use for compaction / parsing: QCAT_HOST_THREADS (the rank launcher sets it to this
// rank's share of the container's CPU quota), else the CPUs of the affinity mask, at most 16
static inline unsigned host_threads() {
    const char* e = getenv("QCAT_HOST_THREADS");
    if (e && atoi(e) > 0) return (unsigned)std::min(atoi(e), 64);
    unsigned n = 0;
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) n = (unsigned)CPU_COUNT(&set);
    if (!n) n = std::max(1u, std::thread::hardware_concurrency());
    return std::min(n, 16u);
}

// temporary device allocation released on every exit path
struct DevTemp {
    void* p = nullptr;
    ~DevTemp() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 1); }
    template <class T> T* as() const { return static_cast<T*>(p); }
};

// ------------------------------------------------------------------------------------------
// kit
// ------------------------------------------------------------------------------------------
constexpr int MAX_DEVICES = 16;



struct KitOnDevice {
    bool ready = false;
    DevKit* kit = nullptr;
    uint8_t* codes = nullptr;
    int32_t* ids = nullptr;
    uint32_t* tables = nullptr;
    char* ascii = nullptr;
    // static-letter kernels compiled for this kit at run time (qcat_kit_attach_code)
    hipModule_t jit_module = nullptr;
    hipFunction_t jit_ad[MAX_T] = {}, jit_am[MAX_T] = {}, jit_bc[MAX_T * 2] = {}, jit_bs[MAX_T * 2] = {};
    hipFunction_t jit_abs[3][MAX_T] = {};          // bit-sliced adapter plans: qj_abs_<t>, qj_absm_<t>, qj_absw_<t>
};

struct qcat_kit {
    HostKit hk;
    std::mutex mu;
    KitOnDevice dev[MAX_DEVICES];
    std::vector<uint8_t> jit_code;                 // code object (gfx950) with qj_ad_<t>, qj_am_<t>, qj_bc_<t*2+s>, qj_bs_<t*2+s>
    bool jit_tpl[MAX_T] = {}, jit_grp[MAX_T * 2] = {}, jit_bsgrp[MAX_T * 2] = {};
    uint64_t serial = 0;                           // never reused: the key of a context's captured graph (ApiGraph)
};

// launch of a run-time generated kernel (kernel id >= QCAT_JIT_BASE): the scan in progress on this
// thread publishes its kit's function table here
static thread_local const KitOnDevice* g_jit = nullptr;

namespace qk {
static inline void jit_launch(int kind, int index, dim3 grid, hipStream_t stream, const void* args) {
    hipFunction_t f = nullptr;
    if (g_jit) f = kind == QCAT_JIT_ADAPTER ? g_jit->jit_ad[index] : (kind == QCAT_JIT_MIDDLE ? g_jit->jit_am[index] :
                   (kind == QCAT_JIT_BITSLICE ? g_jit->jit_bs[index] :
                    (kind >= QCAT_JIT_ABS2 && kind <= QCAT_JIT_ABSW ? g_jit->jit_abs[kind - QCAT_JIT_ABS2][index] : g_jit->jit_bc[index])));
    if (!f) { g_packed_err = "run-time generated kernel missing from the kit's code object"; g_jit_rc = QCAT_ERR_DEVICE; return; }
    void* params[1] = {const_cast<void*>(args)};
    const unsigned threads = kind == QCAT_JIT_BITSLICE ? BS_WAVES * 64 : (kind == QCAT_JIT_ABS2 ? 128u : (kind == QCAT_JIT_ABSM || kind == QCAT_JIT_ABSW ? 256u : PK_WAVES * 64));
    const hipError_t e = hipModuleLaunchKernel(f, grid.x, grid.y, grid.z, threads, 1, 1, 0, stream, params, nullptr);
    if (e != hipSuccess) { g_packed_err = std::string("launch of a run-time generated kernel: ") + hipGetErrorString(e); g_jit_rc = QCAT_ERR_DEVICE; }
}
}  // namespace qk

extern "C" int qcat_kit_create(const qcat_kit_desc* desc, qcat_kit** out) {
    if (!out) return set_err(QCAT_ERR_ARG, "qcat_kit_create: null output pointer");
    qcat_kit* k = new qcat_kit();
    std::string err;
    int rc = kit_prepare(desc, &k->hk, &err);
    if (rc) { delete k; return set_err(rc, err); }
    static_match(&k->hk);
    static std::atomic<uint64_t> next_serial{1};
    k->serial = next_serial.fetch_add(1);
    *out = k;
    return 0;
}

extern "C" int qcat_kit_attach_code_quads(qcat_kit* k, const void* code, uint64_t size,
                                          const int32_t* template_flags, const int32_t* group_flags,
                                          const int32_t* pair_offsets, const int32_t* pair_entries,
                                          const int32_t* quad_offsets, const int32_t* quad_entries);

extern "C" int qcat_kit_attach_code(qcat_kit* k, const void* code, uint64_t size,
                                    const int32_t* template_flags, const int32_t* group_flags,
                                    const int32_t* pair_offsets, const int32_t* pair_entries) {
    return qcat_kit_attach_code_quads(k, code, size, template_flags, group_flags, pair_offsets, pair_entries, nullptr, nullptr);
}

extern "C" int qcat_kit_attach_code_quads(qcat_kit* k, const void* code, uint64_t size,
                                          const int32_t* template_flags, const int32_t* group_flags,
                                          const int32_t* pair_offsets, const int32_t* pair_entries,
                                          const int32_t* quad_offsets, const int32_t* quad_entries) {
    if (quad_offsets && !quad_entries) return set_err(QCAT_ERR_ARG, "qcat_kit_attach_code: quad offsets without quad entries");
    if (!k || !code || !size || !template_flags || !group_flags || !pair_offsets || !pair_entries)
        return set_err(QCAT_ERR_ARG, "qcat_kit_attach_code: null argument");
    std::lock_guard<std::mutex> lock(k->mu);
    for (int d = 0; d < MAX_DEVICES; ++d)
        if (k->dev[d].ready) return set_err(QCAT_ERR_ARG, "qcat_kit_attach_code: the kit is already in use on a device");
    if (!k->jit_code.empty()) return set_err(QCAT_ERR_ARG, "qcat_kit_attach_code: code already attached");
    HostKit& h = k->hk;
    DevKit& d = h.dk;
    const int nsets = d.mode == QCAT_MODE_DUAL ? 2 : 1;
    // validate the pair lists BEFORE anything is bound: the barcode kernel writes its raw scores at the
    // kit barcode indices named here, and k_barcode_select reads one score per barcode of the set
    for (int g = 0; g < 2 * MAX_T; ++g)
        if (pair_offsets[g] < 0 || pair_offsets[g + 1] < pair_offsets[g] ||
            (quad_offsets && (quad_offsets[g] < 0 || quad_offsets[g + 1] < quad_offsets[g])))
            return set_err(QCAT_ERR_ARG, "qcat_kit_attach_code: pair / quad offsets must be non-negative and non-decreasing");
    for (int t = 0; t < d.nt; ++t)
        for (int s = 0; s < nsets; ++s) {
            const DevSet& q = d.tpl[t].sets[s];
            if (!group_flags[t * 2 + s] || !d.barcode_f16 || q.static_kernel >= 0 || q.n <= 0) continue;
            const int32_t* ent = pair_entries + (size_t)pair_offsets[t * 2 + s] * 3;
            const int np = pair_offsets[t * 2 + s + 1] - pair_offsets[t * 2 + s];
            const int nqd = quad_offsets ? quad_offsets[t * 2 + s + 1] - quad_offsets[t * 2 + s] : 0;
            if (np <= 0 && nqd <= 0) continue;
            if (np > q.n || nqd > q.n) return set_err(QCAT_ERR_ARG, "qcat_kit_attach_code: more target pairs than barcodes in a set");
            std::vector<char> seen((size_t)q.n, 0);
            int covered = 0;
            for (int i = 0; i < nqd; ++i) {
                const int32_t* qe = quad_entries + ((size_t)quad_offsets[t * 2 + s] + i) * 5;
                if (qe[0] < 0) return set_err(QCAT_ERR_ARG, "qcat_kit_attach_code: negative quad case");
                for (int m2 = 1; m2 <= 4; ++m2) {
                    const int32_t b = qe[m2];
                    if (b < 0 || b >= q.n) return set_err(QCAT_ERR_ARG, "qcat_kit_attach_code: barcode index outside its set");
                    if (seen[(size_t)b]) return set_err(QCAT_ERR_ARG, "qcat_kit_attach_code: a barcode appears in two target pairs");
                    seen[(size_t)b] = 1; ++covered;
                }
            }
            for (int i = 0; i < np; ++i) {
                if (ent[i * 3] < 0) return set_err(QCAT_ERR_ARG, "qcat_kit_attach_code: negative pair case");
                for (int half = 1; half <= 2; ++half) {
                    const int32_t b = ent[i * 3 + half];
                    if (b == -1 && half == 2) continue;                      // an unpaired target
                    if (b < 0 || b >= q.n) return set_err(QCAT_ERR_ARG, "qcat_kit_attach_code: barcode index outside its set");
                    if (seen[(size_t)b]) return set_err(QCAT_ERR_ARG, "qcat_kit_attach_code: a barcode appears in two target pairs");
                    seen[(size_t)b] = 1; ++covered;
                }
            }
            if (covered != q.n) return set_err(QCAT_ERR_ARG, "qcat_kit_attach_code: the target pairs do not cover every barcode of the set");
        }
    k->jit_code.assign((const uint8_t*)code, (const uint8_t*)code + size);
    for (int t = 0; t < d.nt; ++t) {
        if ((template_flags[t] & 1) && d.adapter_f16 && d.tpl[t].static_kernel < 0) {
            d.tpl[t].static_kernel = QCAT_JIT_BASE + t; k->jit_tpl[t] = true;
            // bits 1..3 of a template flag: the code object also holds bit-sliced adapter plans of this template (qj_abs_<t>
            // two stages, qj_absm_<t> four stages, qj_absw_<t> four wide stages); only kits the path takes at all use them
            if (d.abs_ok) d.tpl[t].abs_jit = (template_flags[t] >> 1) & 7;
        }
        for (int s = 0; s < nsets; ++s) {
            DevSet& q = d.tpl[t].sets[s];
            if (!group_flags[t * 2 + s] || !d.barcode_f16 || q.static_kernel >= 0 || q.n <= 0) continue;
            const int32_t* ent = pair_entries + (size_t)pair_offsets[t * 2 + s] * 3;
            const int np = pair_offsets[t * 2 + s + 1] - pair_offsets[t * 2 + s];
            const int nqd = quad_offsets ? quad_offsets[t * 2 + s + 1] - quad_offsets[t * 2 + s] : 0;
            if (np <= 0 && nqd <= 0) continue;
            q.static_kernel = QCAT_JIT_BASE + t * 2 + s;
            q.n_pairs = np;
            q.case_off = (int32_t)h.ids.size();          // (pair case, barcode of half 0, barcode of half 1) per pair
            h.ids.insert(h.ids.end(), ent, ent + (size_t)np * 3);
            q.n_quads = nqd;
            q.quad_off = (int32_t)h.ids.size();          // (quad case, barcodes a, b, c, d) per quad
            if (nqd > 0) {
                const int32_t* qe = quad_entries + (size_t)quad_offsets[t * 2 + s] * 5;
                h.ids.insert(h.ids.end(), qe, qe + (size_t)nqd * 5);
            }
            k->jit_grp[t * 2 + s] = true;
        }
        // bit 1 of a group flag: the code object also holds qj_bs_<group>, the bit-sliced kernel with this set's
        // letters compiled in (case = barcode index); only sets the bit-sliced path takes at all can use it
        for (int s = 0; s < nsets; ++s) {
            DevSet& q = d.tpl[t].sets[s];
            if (!(group_flags[t * 2 + s] & 2) || q.bs_off < 0 || q.bs_kernel >= 0 || q.n <= 0) continue;
            q.bs_kernel = QCAT_JIT_BASE + t * 2 + s;
            q.bs_case_off = (int32_t)h.ids.size();
            for (int b = 0; b < q.n; ++b) h.ids.push_back(b);
            k->jit_bsgrp[t * 2 + s] = true;
        }
    }
    return 0;
}

extern "C" int qcat_kit_describe(const qcat_kit* k, qcat_kit_info* out) {
    if (!k || !out) return set_err(QCAT_ERR_ARG, "qcat_kit_describe: null argument");
    const DevKit& d = k->hk.dk;
    memset(out, 0, sizeof *out);
    out->packed = (d.fast_ok && packed_supported(d)) ? 1 : 0;
    out->barcode_f16 = d.barcode_f16; out->adapter_f16 = d.adapter_f16;
    out->n_templates = d.nt;
    const int nsets = d.mode == QCAT_MODE_DUAL ? 2 : 1;
    for (int t = 0; t < d.nt; ++t) {
        if (d.tpl[t].static_kernel >= 0) out->n_static_templates++;
        if (d.abs_ok) {
            const int forms = abs_forms(d, t);
            out->bitslice_templates += (forms & 1 ? 1 : 0) + (forms & 2 ? 0x100 : 0) + (forms & 4 ? 0x10000 : 0);
        }
        for (int s = 0; s < nsets; ++s) {
            out->n_groups++;
            if (d.tpl[t].sets[s].static_kernel >= 0) out->n_static_groups++;
            if (d.bs_ok && d.tpl[t].sets[s].bs_off >= 0) out->bitslice_groups += d.tpl[t].sets[s].bs_kernel >= 0 ? 0x10001 : 1;
        }
    }
    return 0;
}

static void kit_device_release(KitOnDevice& kd);

extern "C" void qcat_kit_destroy(qcat_kit* k) {
    if (!k) return;
    int cur = 0;
    (void)hipGetDevice(&cur);
    for (int d = 0; d < MAX_DEVICES; ++d) {
        KitOnDevice& kd = k->dev[d];
        if (!kd.ready) continue;
        (void)hipSetDevice(d);
        kit_device_release(kd);
    }
    (void)hipSetDevice(cur);
    delete k;
}

extern "C" int qcat_kit_count_buckets(const qcat_kit* k) { return k ? k->hk.dk.n_buckets : QCAT_ERR_ARG; }

static void kit_device_release(KitOnDevice& kd) {
    (void)hipFree(kd.kit); (void)hipFree(kd.codes); (void)hipFree(kd.ids); (void)hipFree(kd.tables); (void)hipFree(kd.ascii);
    if (kd.jit_module) (void)hipModuleUnload(kd.jit_module);
    kd = KitOnDevice();
}

static int kit_upload(qcat_kit* k, KitOnDevice& kd) {
    const HostKit& h = k->hk;
    HIPCHK(hipMalloc((void**)&kd.kit, sizeof(DevKit)));
    HIPCHK(hipMalloc((void**)&kd.codes, h.codes.size()));
    HIPCHK(hipMalloc((void**)&kd.ids, h.ids.size() * 4));
    HIPCHK(hipMalloc((void**)&kd.tables, h.tables.size() * 4));
    HIPCHK(hipMalloc((void**)&kd.ascii, std::max<size_t>(1, h.ascii.size())));
    HIPCHK(hipMemcpy(kd.kit, &h.dk, sizeof(DevKit), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(kd.codes, h.codes.data(), h.codes.size(), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(kd.ids, h.ids.data(), h.ids.size() * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(kd.tables, h.tables.data(), h.tables.size() * 4, hipMemcpyHostToDevice));
    if (!h.ascii.empty()) HIPCHK(hipMemcpy(kd.ascii, h.ascii.data(), h.ascii.size(), hipMemcpyHostToDevice));
    if (!k->jit_code.empty()) {
        HIPCHK(hipModuleLoadData(&kd.jit_module, k->jit_code.data()));
        char name[32];
        for (int t = 0; t < h.dk.nt; ++t) {
            if (k->jit_tpl[t]) {
                snprintf(name, sizeof name, "qj_ad_%d", t);
                HIPCHK(hipModuleGetFunction(&kd.jit_ad[t], kd.jit_module, name));
                snprintf(name, sizeof name, "qj_am_%d", t);
                HIPCHK(hipModuleGetFunction(&kd.jit_am[t], kd.jit_module, name));
                static const char* const abs_names[3] = {"qj_abs_%d", "qj_absm_%d", "qj_absw_%d"};
                for (int f = 0; f < 3; ++f) if (h.dk.tpl[t].abs_jit & (1 << f)) {
                    snprintf(name, sizeof name, abs_names[f], t);
                    HIPCHK(hipModuleGetFunction(&kd.jit_abs[f][t], kd.jit_module, name));
                }
            }
            for (int s2 = 0; s2 < 2; ++s2) if (k->jit_grp[t * 2 + s2]) {
                snprintf(name, sizeof name, "qj_bc_%d", t * 2 + s2);
                HIPCHK(hipModuleGetFunction(&kd.jit_bc[t * 2 + s2], kd.jit_module, name));
            }
            for (int s2 = 0; s2 < 2; ++s2) if (k->jit_bsgrp[t * 2 + s2]) {
                snprintf(name, sizeof name, "qj_bs_%d", t * 2 + s2);
                HIPCHK(hipModuleGetFunction(&kd.jit_bs[t * 2 + s2], kd.jit_module, name));
            }
        }
    }
    return 0;
}

static int kit_on_device(qcat_kit* k, int device, KitOnDevice** out) {
    if (device < 0 || device >= MAX_DEVICES) return set_err(QCAT_ERR_ARG, "device index out of range");
    std::lock_guard<std::mutex> lock(k->mu);
    KitOnDevice& kd = k->dev[device];
    if (!kd.ready) {
        const int rc = kit_upload(k, kd);
        if (rc) { kit_device_release(kd); return rc; }     // a failed upload leaves nothing behind: a retry starts clean
        kd.ready = true;
    }
    *out = &kd;
    return 0;
}

// ------------------------------------------------------------------------------------------
// batch (reads resident in device memory)
// ------------------------------------------------------------------------------------------
constexpr size_t BATCH_SLACK = 64;
struct qcat_batch {
    int device = 0;
    uint32_t n_reads = 0;
    uint64_t n_bases = 0;
    uint8_t* bases_alloc = nullptr;   // BATCH_SLACK + n_bases + BATCH_SLACK bytes
    uint8_t* bases = nullptr;      // bases_alloc + BATCH_SLACK: k_pack_windows reads whole aligned dwords
                                   // up to 19 bytes before / after a read's window
    uint64_t* offsets = nullptr;   // n_reads + 1
    uint32_t* true_len = nullptr;  // window-only batches (batch_upload_windows): the reads' real lengths
    bool borrowed = false;         // the device buffers belong to a context (host-buffer calls): destroy frees the shell only
};

extern "C" void qcat_batch_destroy(qcat_batch* b);
// owns a batch under construction: destroyed on every early return, handed over with release()
struct BatchGuard {
    qcat_batch* b;
    explicit BatchGuard(qcat_batch* p) : b(p) {}
    ~BatchGuard() { if (b) qcat_batch_destroy(b); }
    qcat_batch* release() { qcat_batch* t = b; b = nullptr; return t; }
    BatchGuard(const BatchGuard&) = delete;
    BatchGuard& operator=(const BatchGuard&) = delete;
};

// ------------------------------------------------------------------------------------------
// context
// ------------------------------------------------------------------------------------------
constexpr int MAX_TIMED = 12;

struct qcat_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    // per-scan buffers (grown on demand)
    size_t cap_win = 0, cap_wlen = 0, cap_recs = 0, cap_reads = 0, cap_buckets = 0;
    uint8_t* win = nullptr;
    int32_t* wlen = nullptr;
    uint8_t* wspec = nullptr; size_t cap_wspec = 0;     // per read end: the window holds a letter outside A, C, G, T, N
    uint32_t* win2 = nullptr; size_t cap_win2 = 0;      // the windows at two bits per code (plain windows: the bit-sliced kernels' input)
    bool win2_valid = false;                            // ... written by the pack kernel of the windows in `win`
    EndRec* recs = nullptr;
    qcat_result* results = nullptr;
    unsigned long long* counts = nullptr;
    // pinned staging of window-only uploads (qcat_scan_batch): compact bases, offsets, real lengths
    uint8_t* pin_bases = nullptr; size_t cap_pin_bases = 0;
    uint64_t* pin_offsets = nullptr; uint32_t* pin_len = nullptr; size_t cap_pin_reads = 0;
    uint8_t* pin_ret = nullptr; size_t cap_pin_ret = 0;     // what a host-buffer call brings back (records, votes, counts): copies into PINNED
                                                            // memory are one DMA each; into the caller's pageable arrays the runtime stages
                                                            // every one through a copy kernel and a host memcpy (round 6: five of them were
                                                            // 100 us of a 4000-read kit-auto call)
    HostPipeline* pipe = nullptr;                  // chunked host-buffer scans (host_pipeline.inc), created on first use
    std::vector<qcat_ctx*> helpers;                // contexts of the kit-auto file loop's other workers (fastq_host.inc), created on first use
    PackedScratch packed;
    // packed --detect-middle (kernels_middle.inc): M-end sort tables and per-slot arrays
    MidTables* mid_tables = nullptr;
    uint8_t* mid_generic = nullptr; size_t cap_mid_generic = 0;
    uint32_t* mid_slot = nullptr; size_t cap_mid_slot = 0;
    uint32_t* mid_sorted = nullptr; int32_t* mid_len = nullptr; int32_t* mid_fallback = nullptr;
    uint32_t* mid_win2 = nullptr; uint8_t* mid_wspec = nullptr;   // the M-ends' first window at two bits per code + flags (k_mid_windows)
    EndRec* mid_recs = nullptr; AdapterBest* mid_bests = nullptr;
    size_t cap_mid_slots = 0, cap_mid_bests = 0;
    // ... its bit-sliced adapter scan (kernels_abs_mid.inc): per big tile of 2048 slots [rows, padding zone, kit, first row: 4 x tiles]
    // [nonempty, invalid: 2 x 64 x tiles] [cursors: MAX_T], the per-128 flags, letter planes and not-started masks
    uint32_t* absm_tiles = nullptr; size_t cap_absm_tiles = 0;
    uint8_t* absm_need = nullptr; size_t cap_absm_need = 0;
    uint2* absm_planes = nullptr; size_t cap_absm_planes = 0;
    uint32_t* absm_ns = nullptr; size_t cap_absm_ns = 0;
    uint32_t* absm_c2 = nullptr; size_t cap_absm_c2 = 0;          // the batch at two bits per base (with ABSM_C2_SLACK dwords on either side)
    uint8_t* absm_rspec = nullptr; size_t cap_absm_rspec = 0;
    uint4* absm_sinfo = nullptr; size_t cap_absm_sinfo = 0;      // per slot: position / strand, length, special flag
    hipStream_t absm_stream = nullptr; hipEvent_t absm_go = nullptr, absm_done = nullptr;      // the packed batch beside the read ends' kernels
    bool absm_codes_early = false;                                // this scan's k_absmid_codes is in flight on absm_stream
    uint32_t absm_last_big = 0, absm_last_128 = 0;                // big tiles / tiles of 128 slots of the latest scan on that path (0: not taken)
    uint32_t last_n_reads = 0;
    int last_buckets = 0;
    // device staging of the host-buffer calls (qcat_scan_batch / _auto / _debug, qcat_detect_kit): grown on demand and
    // reused -- a 4000-read batch of the reference driver's call shape spent a third of its call in six hipMalloc / hipFree
    uint8_t* hb_bases = nullptr; size_t cap_hb_bases = 0;
    uint64_t* hb_offsets = nullptr; uint32_t* hb_len = nullptr; size_t cap_hb_reads = 0;
    unsigned long long* vote_buf = nullptr;        // per batch of a kit vote: MAX_T counters, MAX_T first voters, then one chosen slot each
    size_t cap_vote_batches = 0;
    // the device work of a kit-auto host-buffer call as a captured graph (scan_batch_auto_impl): two dozen launches (round 5: ~45 on
    // twelve streams) replayed by one hipGraphLaunch when a call has the shape of the one before it
    struct ApiGraph {
        hipGraphExec_t exec = nullptr;
        uint64_t kit = 0, n_bases = 0, gen = 0; uint32_t n_reads = 0, batch_reads = 0;             // what `exec` was captured for
        uint64_t prev_kit = 0, prev_bases = 0, prev_gen = 0; uint32_t prev_reads = 0, prev_batch = 0;   // the shape of the last call
        int failures = 0;                                                         // captures that did not work out (two: never again)
        uint64_t replays = 0;
    } api_graph, scan_graph;                       // kit-auto calls (scan_batch_auto_impl); calls with a named kit (qcat_scan_batch, round 5)
    // the handful-of-reads path (kernels_tiny.inc): per read end the templates' (raw, end) and the barcodes' raw scores
    int32_t* tiny_tpl = nullptr; int16_t* tiny_sc = nullptr; size_t cap_tiny = 0, cap_tiny_ends = 0;
    uint32_t last_tiny_ends = 0;                   // read ends the last scan put on that path (0: another path)
    // --detect-middle: the reads the packed interior scan leaves (long interiors) on the same kernels (k_midw_*)
    uint32_t* midw_list = nullptr; int32_t* midw_tpl = nullptr; EndRec* midw_recs = nullptr; int16_t* midw_sc = nullptr; size_t cap_midw_sc = 0;
    bool midw_ran = false;
    // debug buffers
    int32_t* dbg_tpl = nullptr; size_t cap_dbg_tpl = 0;
    int16_t* dbg_rows = nullptr; size_t cap_dbg_rows = 0;
    // timing: a ring of event sets, one per scan, so that timed scans need no host synchronisation
    // between them; qcat_ctx_last_timing drains the ring
    bool timing = false;
    int force_generic = 0;
    static constexpr int TIME_RING = 64;
    int n_timed = 0;                               // marks of the scan being recorded
    int ring_used = 0;                             // recorded scans not yet drained (<= TIME_RING)
    int ring_marks[TIME_RING];
    const char* timed_name[TIME_RING][MAX_TIMED];
    hipEvent_t evr[TIME_RING][MAX_TIMED + 1];
    hipEvent_t* ev = nullptr;                      // event set of the scan being recorded
    bool ev_ready = false;
};

extern "C" int qcat_ctx_create(int device, qcat_ctx** out) {
    if (!out) return set_err(QCAT_ERR_ARG, "qcat_ctx_create: null output pointer");
    int n = qcat_device_count();
    if (n <= 0) return set_err(QCAT_ERR_DEVICE, "no HIP device visible");
    if (device < 0 || device >= n || device >= MAX_DEVICES) return set_err(QCAT_ERR_ARG, "device index out of range");
    HIPCHK(hipSetDevice(device));
    qcat_ctx* c = new qcat_ctx();
    c->device = device;
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete c; return set_err(QCAT_ERR_DEVICE, std::string("hipStreamCreate: ") + hipGetErrorString(e)); }
    const QOptVal fg = qopt_get(QO_FORCE_GENERIC);
    c->force_generic = (fg && fg.v == 1) ? 1 : 0;
    *out = c;
    return 0;
}

extern "C" void qcat_ctx_destroy(qcat_ctx* c) {
    if (!c) return;
    for (qcat_ctx* h : c->helpers) qcat_ctx_destroy(h);
    c->helpers.clear();
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    if (c->api_graph.exec) (void)hipGraphExecDestroy(c->api_graph.exec);
    if (c->scan_graph.exec) (void)hipGraphExecDestroy(c->scan_graph.exec);
    (void)hipFree(c->win); (void)hipFree(c->wlen); (void)hipFree(c->wspec); (void)hipFree(c->win2); (void)hipFree(c->recs); (void)hipFree(c->results);
    (void)hipFree(c->counts); (void)hipFree(c->dbg_tpl); (void)hipFree(c->dbg_rows); (void)hipFree(c->tiny_tpl); (void)hipFree(c->tiny_sc);
    (void)hipFree(c->midw_list); (void)hipFree(c->midw_tpl); (void)hipFree(c->midw_recs); (void)hipFree(c->midw_sc);
    (void)hipFree(c->hb_bases); (void)hipFree(c->hb_offsets); (void)hipFree(c->hb_len); (void)hipFree(c->vote_buf);
    packed_scratch_free(&c->packed);
    if (c->pin_bases) (void)hipHostFree(c->pin_bases);
    if (c->pin_offsets) (void)hipHostFree(c->pin_offsets);
    if (c->pin_len) (void)hipHostFree(c->pin_len);
    if (c->pin_ret) (void)hipHostFree(c->pin_ret);
    pipeline_free(c->pipe);
    (void)hipFree(c->mid_tables); (void)hipFree(c->mid_generic); (void)hipFree(c->mid_slot); (void)hipFree(c->mid_sorted);
    (void)hipFree(c->mid_len); (void)hipFree(c->mid_fallback); (void)hipFree(c->mid_recs); (void)hipFree(c->mid_bests);
    (void)hipFree(c->mid_win2); (void)hipFree(c->mid_wspec);
    (void)hipFree(c->absm_tiles); (void)hipFree(c->absm_need); (void)hipFree(c->absm_planes); (void)hipFree(c->absm_ns);
    (void)hipFree(c->absm_c2); (void)hipFree(c->absm_rspec); (void)hipFree(c->absm_sinfo);
    if (c->absm_stream) (void)hipStreamDestroy(c->absm_stream);
    if (c->absm_go) (void)hipEventDestroy(c->absm_go);
    if (c->absm_done) (void)hipEventDestroy(c->absm_done);
    if (c->ev_ready) for (int r = 0; r < qcat_ctx::TIME_RING; ++r) for (int i = 0; i <= MAX_TIMED; ++i) (void)hipEventDestroy(c->evr[r][i]);
    (void)hipStreamDestroy(c->stream);
    delete c;
}

extern "C" int qcat_ctx_set_timing(qcat_ctx* c, int enabled) {
    if (!c) return set_err(QCAT_ERR_ARG, "null context");
    HIPCHK(hipSetDevice(c->device));
    if (enabled && !c->ev_ready) {
        for (int r = 0; r < qcat_ctx::TIME_RING; ++r) for (int i = 0; i <= MAX_TIMED; ++i) HIPCHK(hipEventCreate(&c->evr[r][i]));
        c->ev_ready = true;
    }
    c->timing = enabled != 0;
    c->ring_used = 0; c->n_timed = 0; c->ev = nullptr;
    return 0;
}

template <class T>
static int grow(T** p, size_t* cap, size_t need) {
    if (need <= *cap && *p) return 0;
    if (*p) { (void)hipFree(*p); *p = nullptr; }
    size_t n = std::max<size_t>(need, 1);
    hipError_t e = q_malloc((void**)p, n * sizeof(T));
    if (e != hipSuccess) { *cap = 0; return set_err(QCAT_ERR_NOMEM, std::string("hipMalloc: ") + hipGetErrorString(e)); }
    *cap = n;
    return 0;
}

static void mark(qcat_ctx* c, const char* name) {
    if (!c->timing || !c->ev || c->n_timed >= MAX_TIMED) return;
    c->timed_name[c->ring_used - 1][c->n_timed] = name;
    (void)hipEventRecord(c->ev[c->n_timed + 1], c->stream);
    c->n_timed++;
    c->ring_marks[c->ring_used - 1] = c->n_timed;
}

// packed --detect-middle is possible when every template has a generated static-letter adapter
// kernel (all built-in kits) and the kit's slots fit the sort tables
static bool middle_packed_ok(const DevKit& hk) {
    if (!hk.adapter_f16 || hk.n_kit_slots > MID_MAX_KITS || opt_on(QO_MIDDLE_GENERIC)) return false;
    // the interior kernel carries up to PK_ROWS + 63 rows of bias inside a block plus the offset of its
    // last-row keys (kernels_middle.inc): g * ((152 + 63 + 128 + 64) - (160 + 128)) + 1 more than a window
    if (hk.adapter_f16_headroom < hk.gap_open * 119 + 1) return false;
    for (int t = 0; t < hk.nt; ++t) if (hk.tpl[t].static_kernel < 0) return false;
    return true;
}

// the bit-sliced interior adapter scan (kernels_abs_mid.inc, abs_mid_kernels.hip)
extern "C" void qcat_absmid_prepare(void* stream, const void* args, uint32_t* win2, uint8_t* wspec, int what);
extern "C" int qcat_absmid_launch(int id, int waves, unsigned grid, void* stream, const void* args);

// kit slots whose templates all have a two-stage bit-sliced plan (the built-in kits' single-template plans)
static uint32_t absmid_kit_mask(const DevKit& hk) {
    if (!hk.abs_ok) return 0u;
    uint32_t mask = 0u, bad = 0u;
    for (int t = 0; t < hk.nt; ++t) {
        const int ks = hk.tpl[t].kit_slot, sk = hk.tpl[t].static_kernel;
        if (ks < 0 || ks >= 32) continue;
        if (sk >= 0 && sk < QCAT_JIT_BASE && qcat_absmid_launch(sk, 2, 0, nullptr, nullptr)) mask |= 1u << ks; else bad |= 1u << ks;
    }
    return mask & ~bad;
}

// slots of the packed interior scan of n reads, and will its adapter scan run bit-sliced (kit mask != 0)?
static size_t middle_slots(const DevKit& hk, uint32_t n) {
    return ((size_t)2 * n + (size_t)hk.n_kit_slots * MID_CLASSES * PK_TILE + PK_TILE - 1) / PK_TILE * PK_TILE;
}
static uint32_t absmid_wanted(const DevKit& hk, uint32_t n) {
    const QOptVal amin = qopt_get(QO_MIDDLE_ABS_MIN);
    // from 1.25 big tiles of 2048 slots per CU (330 k reads on 256 CUs): below that the binary16 kernel's 3 us per thousand interiors
    // beat a tile's sequential walk (tools/r04_absmid_sizes.sh, profiles/r04_ab_absmid_sizes.txt)
    const size_t min_slots = amin ? (size_t)atoll(amin) : (size_t)abs_cu_count() * 2048 * 5 / 4;
    if (opt_on(QO_MIDDLE_NO_ABS) || middle_slots(hk, n) < min_slots) return 0u;
    return absmid_kit_mask(hk);
}
constexpr size_t ABSM_C2_SLACK_HOST = 1040;                                     // = ABSM_C2_SLACK (kernels_abs_mid.inc)

// QCAT_HIP_MIDDLE_ABS_EARLY=1 (A/B switch, off by default): the batch at two bits per base (k_absmid_codes) at the start of a
// --detect-middle scan on a stream of its own, beside the read ends' kernels -- it needs the reads only; middle_packed waits
// for absm_done.  Measured at 1 M reads: the interior phase loses the kernel's 0.12 ms, k_pack_windows and the plane kernel of
// the read ends -- bound by memory themselves -- gain 0.08 + 0.06 ms: 5.14 ms per step either way.
static int absmid_codes_early(qcat_ctx* c, const DevKit& hk, const qcat_batch* b, uint32_t n) {
    c->absm_codes_early = false;
    if (!((opt_is_set(QO_MIDDLE_ABS_EARLY) && opt_val(QO_MIDDLE_ABS_EARLY, 0) == 1))) return 0;
    if (!absmid_wanted(hk, n)) return 0;
    int rc;
    if (!c->absm_stream) {
        HIPCHK(hipStreamCreateWithFlags(&c->absm_stream, hipStreamNonBlocking));
        HIPCHK(hipEventCreateWithFlags(&c->absm_go, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&c->absm_done, hipEventDisableTiming));
    }
    if ((rc = grow(&c->absm_c2, &c->cap_absm_c2, (size_t)(b->n_bases / 16) + 4 + 2 * ABSM_C2_SLACK_HOST))) return rc;
    if ((rc = grow(&c->absm_rspec, &c->cap_absm_rspec, (size_t)n + 1))) return rc;
    AbsMidArgs am{};
    am.bases = b->bases; am.offsets = b->offsets; am.max_align = hk.max_align; am.n_bases = b->n_bases; am.n_reads = n;
    am.c2 = c->absm_c2 + ABSM_C2_SLACK_HOST; am.rspec = c->absm_rspec;
    HIPCHK(hipEventRecord(c->absm_go, c->stream));                 // (after whatever the stream holds: the previous scan's readers of c2)
    HIPCHK(hipStreamWaitEvent(c->absm_stream, c->absm_go, 0));
    HIPCHK(hipMemsetAsync(c->absm_rspec, 0, (size_t)n + 1, c->absm_stream));
    qcat_absmid_prepare(c->absm_stream, &am, nullptr, nullptr, 1);
    HIPCHK(hipEventRecord(c->absm_done, c->absm_stream));
    c->absm_codes_early = true;
    return 0;
}

// the interior scan of every called read on the packed kernels (kernels_middle.inc); leaves
// c->mid_generic[r] = 1 for reads it could not take (interior longer than the length classes)
static int middle_packed(qcat_ctx* c, KitPtrs kp, const DevKit& hk, const qcat_batch* b, uint32_t n) {
    hipStream_t st = c->stream;
    int rc;
    const size_t slots = middle_slots(hk, n);
    if (slots >= (1ull << 31)) return set_err(QCAT_ERR_UNSUPPORTED, "batch too large for the packed interior scan");
    if (!c->mid_tables) HIPCHK(hipMalloc((void**)&c->mid_tables, sizeof(MidTables)));
    if ((rc = grow(&c->mid_generic, &c->cap_mid_generic, (size_t)n))) return rc;
    if ((rc = grow(&c->mid_slot, &c->cap_mid_slot, (size_t)n))) return rc;
    if (slots > c->cap_mid_slots) {
        (void)hipFree(c->mid_sorted); (void)hipFree(c->mid_len); (void)hipFree(c->mid_fallback); (void)hipFree(c->mid_recs);
        (void)hipFree(c->mid_win2); (void)hipFree(c->mid_wspec);
        c->mid_sorted = nullptr; c->mid_len = nullptr; c->mid_fallback = nullptr; c->mid_recs = nullptr; c->cap_mid_slots = 0;
        c->mid_win2 = nullptr; c->mid_wspec = nullptr;
        HIPCHK(hipMalloc((void**)&c->mid_win2, (slots * WIN2_WORDS + 64) * 4));
        HIPCHK(hipMalloc((void**)&c->mid_wspec, slots + 4));
        HIPCHK(hipMalloc((void**)&c->mid_sorted, slots * 4));
        HIPCHK(hipMalloc((void**)&c->mid_len, slots * 4));
        HIPCHK(hipMalloc((void**)&c->mid_fallback, slots * 4));
        HIPCHK(hipMalloc((void**)&c->mid_recs, slots * sizeof(EndRec)));
        c->cap_mid_slots = slots;
    }
    if ((rc = grow(&c->mid_bests, &c->cap_mid_bests, slots * (size_t)hk.nt))) return rc;
    // the fills of the interior scan leave as two k_fill_multi launches (round 5; thirteen memsets of 5-7 us each before):
    // the slot tables here, the job tables / sorted list / redo count / letter flags / tile cursors before the adapter phase
    const bool merge_fills = !opt_on(QO_NO_FILL_MERGE);
    g_fill.n = 0; g_fill_defer = merge_fills;
    HIPCHK(packed_fill(c->mid_tables, 0, sizeof(MidTables), st));
    HIPCHK(packed_fill(c->mid_slot, 0xFF, (size_t)n * 4, st));
    HIPCHK(packed_fill(c->mid_sorted, 0xFF, slots * 4, st));
    HIPCHK(packed_fill(c->mid_len, 0, slots * 4, st));
    HIPCHK(packed_fill(c->mid_fallback, 0, slots * 4, st));
    HIPCHK(packed_fill_flush(st));
    const uint32_t rblocks = (uint32_t)std::min<uint64_t>(((uint64_t)n + 255) / 256, 2048);
    hipLaunchKernelGGL(k_mid_count, dim3(rblocks), dim3(256), 0, st, kp.kit, b->offsets, n, c->results, c->mid_tables, c->mid_generic);
    hipLaunchKernelGGL(k_mid_offsets, dim3(1), dim3(1024), 0, st, c->mid_tables);
    hipLaunchKernelGGL(k_mid_scatter, dim3((n + 256 * MID_SCAT - 1) / (256 * MID_SCAT)), dim3(256), 0, st,
                       kp.kit, b->offsets, n, c->results, c->mid_tables, c->mid_sorted, c->mid_len, c->mid_fallback, c->mid_slot);
    // adapter phase: one launch per template (tiles of other kits record "did not compete"), the
    // launches of a scan run concurrently
    PackedScratch* sc = &c->packed;
    // round 4: the interiors' first window at two bits per code, so that their whole-window barcode jobs -- nearly all of
    // them -- run on the bit-sliced barcode kernels (QCAT_HIP_MIDDLE_NO_BITSLICE=1: binary16 kernels as before)
    const bool mid_bs = !opt_on(QO_MIDDLE_NO_BITSLICE);
    // (with the bit-sliced adapter scan below the windows come from its packed batch instead: k_absmid_windows.  k_mid_windows beside
    //  the adapter kernels was measured and lost: those fill the register files, each of their waves walks ONE tile, and a wave that
    //  starts late is the kernel's tail; beside the preparation chain -- kernels that wait for memory like itself -- it gained 0.02 ms)
    auto launch_mid_windows = [&](hipStream_t q) {
        const uint64_t wthreads = (uint64_t)slots * 16;           // (sixteen lanes per slot; every slot's flag byte is stored, no fill)
        hipLaunchKernelGGL(k_mid_windows, dim3((uint32_t)((wthreads + 255) / 256)), dim3(256), 0, q, b->bases, b->offsets, hk.max_align,
                           c->mid_sorted, (uint32_t)slots, c->mid_win2, c->mid_wspec);
    };
    sc->wspec = mid_bs ? c->mid_wspec : nullptr;          // (null: no letter flags, no bit-sliced classes)
    sc->win2 = mid_bs ? c->mid_win2 : nullptr;
    sc->win = c->win;                                     // (never read: the job regions come from win2, `lazy`)
    sc->lazy = mid_bs;
    sc->slim = false;                                     // ... and the barcode results stay in the interior's own records
    g_fill.n = 0; g_fill_defer = merge_fills;                  // (flushed in front of the adapter phase below; every return before drops the list)
    struct FillGuard { ~FillGuard() { g_fill_defer = false; g_fill.n = 0; } } fill_guard;
    if ((rc = packed_prepare(st, hk, (uint32_t)slots, sc))) return set_err(rc, packed_last_error());
    const uint32_t tiles = (uint32_t)(slots / PK_TILE);
    // round 4: the adapter scan of the interiors in bit-sliced form (kernels_abs_mid.inc) from QCAT_HIP_MIDDLE_ABS_MIN slots
    // (default: one big tile of 2048 slots per CU); QCAT_HIP_MIDDLE_NO_ABS=1: the binary16 kernel for every tile as before.
    // The binary16 kernel then only takes the tiles of 128 slots flagged in absm_need.
    AbsMidArgs am{};
    bool use_absm = false;
    c->absm_last_big = 0; c->absm_last_128 = 0;
    {
        const uint32_t kmask = absmid_wanted(hk, n);
        if (kmask) {
            const uint32_t big = (uint32_t)((slots + 2047) / 2048);
            // rows of all big tiles together: every tile as long as its longest interior -- the mean of the interiors plus
            // the width of the length classes a tile spans; a tile beyond the room falls back to the binary16 kernel
            const QOptVal rcap = qopt_get(QO_MIDDLE_ABS_ROWS);         // (tests: a plane buffer that is too small)
            size_t rows = rcap ? (size_t)atoll(rcap)
                               : std::min<size_t>((size_t)1 << 30, (size_t)(2 * b->n_bases / 2048) * 5 / 4 + (size_t)big * 128 + 16384 + 64);
            if (!rcap && rows * 64 > c->cap_absm_planes) {
                // planes + not-started masks are 768 bytes per row (~0.9 bytes per base of the batch): never more than a quarter
                // of the device memory that is free right now -- the big tiles beyond the room fall back to the binary16 kernel
                // per tile (k_absmid_scan: row_cap), which needs none of it
                size_t free_b = 0, total_b = 0;
                if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) rows = std::min(rows, std::max<size_t>(16384, free_b / 4 / 768));
            }
            const size_t tw = (size_t)big * (4 + 128) + MAX_T;
            if ((rc = grow(&c->absm_tiles, &c->cap_absm_tiles, tw))) return rc;
            if ((rc = grow(&c->absm_need, &c->cap_absm_need, (size_t)tiles + 16))) return rc;
            if ((rc = grow(&c->absm_planes, &c->cap_absm_planes, rows * 64))) return rc;
            if ((rc = grow(&c->absm_ns, &c->cap_absm_ns, rows * 64))) return rc;
            constexpr size_t C2_SLACK = ABSM_C2_SLACK_HOST;
            const bool early = c->absm_codes_early;                             // (absmid_codes_early: the packed batch is on its way)
            c->absm_codes_early = false;
            if (!early) {
                if ((rc = grow(&c->absm_c2, &c->cap_absm_c2, (size_t)(b->n_bases / 16) + 4 + 2 * C2_SLACK))) return rc;
                if ((rc = grow(&c->absm_rspec, &c->cap_absm_rspec, (size_t)n + 1))) return rc;
            }
            if ((rc = grow(&c->absm_sinfo, &c->cap_absm_sinfo, slots))) return rc;
            uint32_t* w = c->absm_tiles;
            am.bases = b->bases; am.offsets = b->offsets; am.msorted = c->mid_sorted; am.mlen = c->mid_len; am.mt = c->mid_tables;
            am.max_align = hk.max_align; am.n_tiles = big; am.slot_cap = (uint32_t)slots; am.nt = hk.nt; am.kit_mask = kmask;
            am.t_rows = (int32_t*)w; am.t_pz = (int32_t*)(w + big); am.t_kit = (int32_t*)(w + 2 * (size_t)big); am.t_off = w + 3 * (size_t)big;
            am.nonempty = w + 4 * (size_t)big; am.invalid = w + 4 * (size_t)big + 64 * (size_t)big;
            am.cursor = w + (size_t)big * (4 + 128);
            am.c2 = c->absm_c2 + C2_SLACK; am.rspec = c->absm_rspec; am.sinfo = c->absm_sinfo; am.n_bases = b->n_bases; am.n_reads = n;
            if (early) HIPCHK(hipStreamWaitEvent(st, c->absm_done, 0));
            else HIPCHK(packed_fill(c->absm_rspec, 0, (size_t)n + 1, st));
            am.need128 = c->absm_need; am.planes = c->absm_planes; am.ns = c->absm_ns; am.row_cap = (uint32_t)rows;
            am.bests = c->mid_bests; am.tpl = -1; am.den = 0; am.kit_slot = -1;
            // issue priority of the row loops rotated every two rows by workgroup parity (abs_setprio; two waves of different
            // launches share a SIMD): 4.79 against 4.91 ms per step at 1 M reads (tools/r04_absmid_prio.sh); QCAT_HIP_MIDDLE_ABS_PRIO
            const QOptVal pr = opt_is_set(QO_MIDDLE_ABS_PRIO) ? qopt_get(QO_MIDDLE_ABS_PRIO) : qopt_get(QO_ABS_PRIO);
            am.prio = pr ? atoi(pr) : 2;
            am.r1_scalar = hk.r1_scalar;
            HIPCHK(packed_fill(am.cursor, 0, MAX_T * 4, st));
            HIPCHK(packed_fill_flush(st));
            // the M-ends' first windows (k_mid_windows) come from the packed batch too: QCAT_HIP_MIDDLE_ABS_WINDOWS=0: from the reads, beside
            const bool c2win = mid_bs && !((opt_is_set(QO_MIDDLE_ABS_WINDOWS) && opt_val(QO_MIDDLE_ABS_WINDOWS, 0) == 0));
            fork_join(sc, st, (mid_bs && !c2win) ? 2 : 1, [&](int i, hipStream_t q) {
                if (i == 0) qcat_absmid_prepare(q, &am, c2win ? c->mid_win2 : nullptr, c2win ? c->mid_wspec : nullptr, early ? 2 : 3); else launch_mid_windows(q);
            });
            use_absm = true;
            c->absm_last_big = big; c->absm_last_128 = tiles;
        }
    }
    HIPCHK(packed_fill_flush(st));                                 // (no bit-sliced interior scan: the job tables' fills leave here)
    int absm_side = 0;                                             // templates on the bit-sliced path: launches side by side
    if (use_absm) for (int t = 0; t < hk.nt; ++t) if ((am.kit_mask >> hk.tpl[t].kit_slot) & 1u) ++absm_side;
    if (mid_bs && !use_absm) launch_mid_windows(st);
    fork_join(sc, st, hk.nt, [&](int t, hipStream_t q) {
        if (use_absm && ((am.kit_mask >> hk.tpl[t].kit_slot) & 1u)) {
            AbsMidArgs at = am;
            at.bests = c->mid_bests + (size_t)t * slots; at.tpl = t; at.den = hk.tpl[t].den; at.kit_slot = hk.tpl[t].kit_slot;
            at.cursor = am.cursor + t;
            // persistent two-wave workgroups: four per CU (two waves per SIMD) shared by the templates that run side by side --
            // two launches of four per CU each do not fit the register file together, and the second one then runs after
            // the first (1.2 + 0.6 ms at 1 M reads against ~1.0 ms side by side).  QCAT_HIP_MIDDLE_ABS_WGS=<per CU and launch>
            // A template of up to 46 columns walks a tile on ONE wave (k_adapter_mid1: eight per CU); QCAT_HIP_MIDDLE_ABS_ONE_WAVE=0 / 1: pipeline / one wave whatever the size.
            const QOptVal wg = qopt_get(QO_MIDDLE_ABS_WGS);
            const QOptVal ow = qopt_get(QO_MIDDLE_ABS_ONE_WAVE);
            const int sk = hk.tpl[t].static_kernel;
            // one wave per tile from 2.75 big tiles per CU (720 k reads): with fewer tiles than wave slots a tile's walk is the kernel,
            // and the pipeline's two waves halve it (400 k reads: interior phase 1.95 against 2.29 ms; 800 k: 2.96 against 2.87)
            const bool big_enough = slots >= (size_t)abs_cu_count() * 2048 * 11 / 4;
            const bool one = (ow ? atoi(ow) != 0 : big_enough) && qcat_absmid_launch(sk, 1, 0, nullptr, nullptr) != 0;
            const int per_cu = wg ? atoi(wg) : std::max(1, (one ? 8 : 4) / std::max(1, absm_side));
            const unsigned grid = (unsigned)std::min<uint32_t>(am.n_tiles, (uint32_t)(abs_cu_count() * per_cu));
            (void)qcat_absmid_launch(sk, one ? 1 : 2, grid, q, &at);
        }
        MiddleAdapterArgs ma{kp, b->bases, b->offsets, c->mid_sorted, c->mid_len, c->mid_tables,
                             c->mid_bests + (size_t)t * slots, t, use_absm ? c->absm_need : nullptr};
        launch_adapter_middle(hk.tpl[t].static_kernel, dim3(tiles), q, ma);
    });
    {
        const uint32_t fblocks = (uint32_t)std::min<uint64_t>((slots + 255) / 256, 2048);
        hipLaunchKernelGGL(k_adapter_finish, dim3(fblocks), dim3(256), 0, st, kp.kit, c->mid_len, (uint32_t)slots,
                           c->mid_bests, hk.nt, c->mid_recs, sc->jt, (const int32_t*)c->mid_fallback, -1, (const uint8_t*)sc->wspec, sc->jobinfo, (int2*)nullptr, (uint32_t*)nullptr);
    }
    const int nsets = hk.mode == QCAT_MODE_DUAL ? 2 : 1;
    rc = packed_barcode(st, kp, hk, (uint32_t)slots, c->mid_recs, sc, [&](uint32_t max_tiles, const BsPlan* taken, hipStream_t gq) {
        const uint64_t gthreads = (uint64_t)max_tiles * 64 * (WIN_STRIDE / 16);
        hipLaunchKernelGGL(k_mid_gather, dim3((uint32_t)((gthreads + 255) / 256)), dim3(256), 0, gq,
                           b->bases, b->offsets, hk.max_align, c->mid_sorted, c->mid_recs, sc->sorted, sc->jt, nsets,
                           sc->tilebuf, sc->tmeta, taken);
    }, (int16_t*)nullptr, 0u, [](const char*) {});
    if (!rc) rc = jit_take_error();
    if (rc) return set_err(rc, packed_last_error());
    hipLaunchKernelGGL(k_mid_finalize, dim3((n + 255) / 256), dim3(256), 0, st, kp.kit, c->mid_recs, c->mid_slot, n, c->results, c->mid_generic);
    HIPCHK(hipGetLastError());
    return 0;
}

// scratch of the one-wave-per-alignment kernels (kernels_tiny.inc) for `n_ends` queries of a kit whose largest set has `maxb` barcodes
static int tiny_buffers(qcat_ctx* c, size_t n_ends, int maxb) {
    const size_t need = n_ends * 2 * (size_t)maxb;
    if (need > c->cap_tiny || n_ends > c->cap_tiny_ends || !c->tiny_tpl) {
        (void)hipFree(c->tiny_tpl); (void)hipFree(c->tiny_sc); c->tiny_tpl = nullptr; c->tiny_sc = nullptr; c->cap_tiny = 0; c->cap_tiny_ends = 0;
        const size_t cap = std::max<size_t>(need, (size_t)TINY_MAX_WAVES * 2), cap_ends = std::max<size_t>(n_ends, 4096);
        HIPCHK(q_malloc((void**)&c->tiny_sc, cap * sizeof(int16_t)));
        HIPCHK(q_malloc((void**)&c->tiny_tpl, cap_ends * MAX_T * 2 * sizeof(int32_t)));
        c->cap_tiny = cap; c->cap_tiny_ends = cap_ends;
    }
    return 0;
}

// core: scan a resident batch.  dbg: optional debug buffers sized by the caller.
static int scan_resident_impl(qcat_ctx* c, qcat_kit* kit, const qcat_batch* b, bool debug, uint32_t row_stride,
                              bool adapter_only = false, int resume_kit_mask = -1, bool keep_counts = false) {
    if (!c || !kit || !b) return set_err(QCAT_ERR_ARG, "null argument");
    if (b->device != c->device) return set_err(QCAT_ERR_ARG, "batch lives on another device than the context");
    HIPCHK(hipSetDevice(c->device));
    KitOnDevice* kd = nullptr;
    int rc = kit_on_device(kit, c->device, &kd);
    if (rc) return rc;
    const DevKit& hk = kit->hk.dk;
    const int ends = hk.ends == QCAT_ENDS_5P ? 1 : 2;
    const uint32_t n = b->n_reads;
    const size_t n_ends = (size_t)n * ends;
    if (n_ends >= (1ull << 31)) return set_err(QCAT_ERR_UNSUPPORTED, "batch too large (>= 2^31 read ends)");

    if ((rc = grow(&c->win, &c->cap_win, n_ends * WIN_STRIDE + 512))) return rc;    // + slack: k_job_gather reads whole dwords, k_bs_barcode 84 bytes from any region start
    if ((rc = grow(&c->wlen, &c->cap_wlen, n_ends))) return rc;
    if ((rc = grow(&c->wspec, &c->cap_wspec, n_ends))) return rc;
    // + slack: readers take whole dwords past a region's end -- and, round 5, one dword in FRONT of the first window (a front-padded
    // unit of the bit-sliced barcode kernels fetches from up to BS_PAD_ROWS bases before a region): the windows start WIN2_FRONT words in
    if ((rc = grow(&c->win2, &c->cap_win2, n_ends * WIN2_WORDS + 64 + WIN2_FRONT))) return rc;
    if ((rc = grow(&c->recs, &c->cap_recs, n_ends))) return rc;
    if ((rc = grow(&c->results, &c->cap_reads, (size_t)n))) return rc;
    if ((rc = grow(&c->counts, &c->cap_buckets, (size_t)hk.n_buckets))) return rc;
    if (debug) {
        if ((rc = grow(&c->dbg_tpl, &c->cap_dbg_tpl, n_ends * 2 * MAX_T))) return rc;
        if (row_stride && (rc = grow(&c->dbg_rows, &c->cap_dbg_rows, n_ends * 2 * row_stride))) return rc;
        HIPCHK(hipMemsetAsync(c->dbg_tpl, 0, n_ends * 2 * MAX_T * 4, c->stream));
        if (row_stride) HIPCHK(hipMemsetD16Async((hipDeviceptr_t)c->dbg_rows, (unsigned short)0x8000, n_ends * 2 * row_stride, c->stream));
    }
    c->last_n_reads = n;
    c->last_buckets = hk.n_buckets;
    c->n_timed = 0;
    c->ev = nullptr;
    if (c->timing) {                               // next slot of the ring; a full ring restarts (oldest scans dropped)
        if (c->ring_used >= qcat_ctx::TIME_RING) c->ring_used = 0;
        c->ring_marks[c->ring_used] = 0;
        c->ev = c->evr[c->ring_used++];
    }
    // the fills of a scan leave as ONE launch in front of the pack kernel (k_fill_multi): count vector, letter flags, and --
    // with the job tables and the adapter tile flags sized here instead of after the pack kernel -- theirs too
    const bool packed_kit = hk.fast_ok && !c->force_generic && packed_supported(hk);
    bool use_packed = packed_kit;
    // a handful of read ends (detect_barcode on one read, a few reads of a test): one wave per alignment, kernels_tiny.inc
    // (QCAT_HIP_TINY_MAX_ENDS: the largest batch that goes there, 0: none; QCAT_HIP_NO_TINY=1)
    // the path pays while its waves -- one per (read end, template) plus one per (read end, set, barcode) -- fit the chip a few
    // times over: up to TINY_MAX_WAVES of them (tools/tiny_crossover.py: PBC096 wins to ~180 reads, NBD104 to ~700)
    const QOptVal tiny_env = qopt_get(QO_TINY_MAX_ENDS);
    int tiny_maxb = 1;
    for (int t = 0; t < hk.nt; ++t) for (int s2 = 0; s2 < 2; ++s2) tiny_maxb = std::max(tiny_maxb, (int)hk.tpl[t].sets[s2].n);
    const uint64_t tiny_waves = n_ends * (uint64_t)(hk.nt + tiny_maxb * (hk.mode == QCAT_MODE_DUAL ? 2 : 1));
    const bool tiny_fits = tiny_env ? n_ends <= (uint64_t)std::max<long long>(0, atoll(tiny_env)) : tiny_waves <= (uint64_t)TINY_MAX_WAVES;
    const bool tiny = n_ends > 0 && n_ends <= 4096 && tiny_fits && !opt_on(QO_NO_TINY) && hk.mode != QCAT_MODE_SIMPLE && resume_kit_mask < 0 &&
                      !adapter_only && !c->force_generic && hk.gap_open == hk.gap_extend;
    c->last_tiny_ends = tiny ? (uint32_t)n_ends : 0;
    if (tiny) {
        if ((rc = tiny_buffers(c, (size_t)n_ends, tiny_maxb))) return rc;
        use_packed = false;                        // (no job tables, no lazy windows: the tiny kernels read byte windows)
    }
    g_fill_defer = n != 0 && !opt_on(QO_NO_FILL_MERGE);
    if (!keep_counts) HIPCHK(packed_fill(c->counts, 0, (size_t)hk.n_buckets * 8, c->stream));
    if (n == 0) { HIPCHK(packed_fill_flush(c->stream)); return 0; }
    if (c->timing) HIPCHK(hipEventRecord(c->ev[0], c->stream));   // (an empty batch returned above: its slot holds no marks)
    c->absm_codes_early = false;
    if (hk.scan_middle && !adapter_only && packed_kit && middle_packed_ok(hk) && (rc = absmid_codes_early(c, hk, b, n))) return rc;

    KitPtrs kp{kd->kit, kd->codes, kd->ids, kd->tables};
    g_jit = kd;
    c->packed.prepared = false; c->packed.abs_zeroed = 0; c->packed.redo_zeroed = false;
    if (resume_kit_mask < 0) {
        uint64_t threads = (uint64_t)n_ends * (WIN_STRIDE / 16);
        uint32_t blocks = (uint32_t)((threads + 255) / 256);
        HIPCHK(packed_fill(c->wspec, 0, n_ends, c->stream));
        c->packed.abs_ready = 0;
        if (g_fill_defer && use_packed && hk.mode != QCAT_MODE_SIMPLE) {
            c->packed.slim = use_packed && !debug && !opt_on(QO_NO_SLIM);     // (packed_prepare sizes the slim buffers)
            if ((rc = packed_prepare(c->stream, hk, (uint32_t)n_ends, &c->packed))) { g_fill_defer = false; g_fill.n = 0; return set_err(rc, packed_last_error()); }
            c->packed.prepared = true;
            if (!opt_on(QO_PACK_PLANES) && packed_abs_wanted(hk, (uint32_t)n_ends, true) &&
                (rc = packed_abs_buffers(c->stream, hk, (uint32_t)n_ends, &c->packed))) { g_fill_defer = false; g_fill.n = 0; return set_err(rc, packed_last_error()); }
        }
        HIPCHK(packed_fill_flush(c->stream));
        if (use_packed && hk.mode != QCAT_MODE_SIMPLE && opt_on(QO_PACK_PLANES) && packed_abs_wanted(hk, (uint32_t)n_ends, true)) {
            // (A/B switch, off by default) the windows and their letter planes in one pass (kernels_abs.inc: k_pack_planes).
            // Measured and dropped: 5.1 ms against 1.67 + 0.85 ms for k_pack_windows + k_abs_planes on config 3 -- a wave that
            // walks eight slices of 640 items serially is latency-bound on the offsets -> bases load chains, where
            // k_pack_windows has one thread per item (profiles/r03_ab_pack_planes.json)
            if ((rc = packed_abs_buffers(c->stream, hk, (uint32_t)n_ends, &c->packed))) return set_err(rc, packed_last_error());
            const uint32_t tiles = ((uint32_t)n_ends + ABS_TILE - 1) / ABS_TILE;
            uint32_t* cursor = reinterpret_cast<uint32_t*>(c->packed.abs_flags);
            uint32_t* tile_any = cursor + MAX_T;
            qcat_abs_launch_pack_planes(tiles, c->stream, b->bases, b->offsets, n, ends, c->win, c->wlen, c->wspec, (uint32_t)n_ends, hk.max_align,
                                        c->packed.abs_planes, c->packed.abs_valid, reinterpret_cast<uint8_t*>(tile_any + tiles), tile_any);
            c->packed.abs_ready = (uint32_t)n_ends;
            c->win2_valid = false;                       // (that kernel does not make the two-bit copy)
            c->packed.lazy = false;
        } else {
            c->win2_valid = true;
            // lazy byte windows (round 4): a kit whose every template runs a generated kernel reads plain windows at two
            // bits per code everywhere (dev_window16), so the byte windows -- 3.3 of the pack kernel's 4.1 GB of stores per
            // 10 M reads -- are written for the flagged ends only (k_expand_special; QCAT_HIP_EAGER_BYTES=1: all of them)
            bool lazy = use_packed && hk.mode != QCAT_MODE_SIMPLE && hk.adapter_f16 && !opt_on(QO_NO_STATIC) &&
                        !opt_on(QO_NO_STATIC_ADAPTER) && !opt_on(QO_EAGER_BYTES);
            for (int t = 0; t < hk.nt && lazy; ++t) lazy = hk.tpl[t].static_kernel >= 0;
            c->packed.lazy = lazy;
            hipLaunchKernelGGL(k_pack_windows, dim3(blocks), dim3(256), 0, c->stream,
                               b->bases, b->offsets, n, ends, hk.max_align, c->win, c->wlen, c->wspec, c->win2 + WIN2_FRONT, lazy ? 1 : 0);
            if (lazy) hipLaunchKernelGGL(k_expand_special, dim3((uint32_t)((n_ends + 255) / 256)), dim3(256), 0, c->stream,
                                         b->bases, b->offsets, n, ends, hk.max_align, c->wspec, c->win);
        }
        mark(c, "k_pack_windows");
    }
    if (resume_kit_mask >= 0 && g_fill_defer && use_packed && c->packed.slices_single && hk.mode != QCAT_MODE_SIMPLE) {
        // (a resumed scan -- the second pass of a kit-auto batch: its job tables' fills with the count vector's, one launch)
        c->packed.slim = use_packed && !debug && !opt_on(QO_NO_SLIM);
        if ((rc = packed_prepare(c->stream, hk, (uint32_t)n_ends, &c->packed))) { g_fill_defer = false; g_fill.n = 0; return set_err(rc, packed_last_error()); }
        c->packed.prepared = true;
    }
    HIPCHK(packed_fill_flush(c->stream));               // (a resumed scan: the count vector's fill)
    if (hk.mode == QCAT_MODE_SIMPLE && adapter_only) return set_err(QCAT_ERR_ARG, "simple mode has no adapter templates to vote with");
    if (resume_kit_mask >= 0 && !(use_packed && c->packed.slices_single))
        return set_err(QCAT_ERR_UNSUPPORTED, "the adapter pass of this kit cannot be resumed per kit (table or general kernels)");
    c->packed.wspec = c->wspec; c->packed.win = c->win; c->packed.win2 = c->win2_valid ? c->win2 + WIN2_FRONT : nullptr;
    // packed barcode results (kernels_bitslice.inc: k_bs_select_ordered) instead of 8 bytes scattered into every record;
    // debug scans keep the records complete for the traces
    const bool slim = use_packed && !debug && hk.mode != QCAT_MODE_SIMPLE && !opt_on(QO_NO_SLIM);
    c->packed.slim = slim;
    if (hk.mode == QCAT_MODE_SIMPLE) {
        uint32_t blocks = (uint32_t)((n_ends + GEN_THREADS - 1) / GEN_THREADS);
        hipLaunchKernelGGL(k_scan_simple, dim3(blocks), dim3(GEN_THREADS), 0, c->stream, kp, c->win, c->wlen, (uint32_t)n_ends, c->recs,
                           (debug && row_stride) ? c->dbg_rows : nullptr, row_stride);
        mark(c, "k_scan_simple");
    } else if (use_packed) {
        rc = packed_scan(c->stream, kp, hk, c->win, c->wlen, (uint32_t)n_ends, c->recs, &c->packed,
                         debug ? c->dbg_tpl : nullptr, (debug && row_stride) ? c->dbg_rows : nullptr, row_stride,
                         [&](const char* nm) { mark(c, nm); }, adapter_only, resume_kit_mask);
        if (rc) return set_err(rc, packed_last_error());
    } else if (tiny) {
        TinyArgs ta{kp, c->win, c->wlen, nullptr, nullptr, (uint32_t)n_ends, c->recs, c->tiny_tpl, c->tiny_sc, (uint32_t)tiny_maxb,
                    debug ? c->dbg_tpl : nullptr, (debug && row_stride) ? c->dbg_rows : nullptr, row_stride, 0};
        hipLaunchKernelGGL(k_tiny_adapter, dim3((uint32_t)n_ends * (uint32_t)hk.nt), dim3(64), 0, c->stream, ta);
        hipLaunchKernelGGL(k_tiny_decide, dim3((uint32_t)((n_ends + 63) / 64)), dim3(64), 0, c->stream, ta);
        hipLaunchKernelGGL(k_tiny_barcode, dim3((uint32_t)tiny_maxb, (uint32_t)n_ends * 2), dim3(64), 0, c->stream, ta);
        hipLaunchKernelGGL(k_tiny_select, dim3((uint32_t)n_ends * 2), dim3(64), 0, c->stream, ta);
        mark(c, "k_scan_tiny");
    } else {
        uint32_t blocks = (uint32_t)((n_ends + GEN_THREADS - 1) / GEN_THREADS);
        hipLaunchKernelGGL(k_scan_generic, dim3(blocks), dim3(GEN_THREADS), 0, c->stream,
                           kp, c->win, c->wlen, (uint32_t)n_ends, c->recs,
                           debug ? c->dbg_tpl : nullptr, (debug && row_stride) ? c->dbg_rows : nullptr, row_stride,
                           adapter_only ? 1 : 0);
        mark(c, "k_scan_generic");
    }
    if (!adapter_only) {
        // every block zeroes and flushes an LDS histogram of n_buckets entries with global atomics on the same few counters:
        // fewer, fatter blocks -- 1024 (measured, tools/r04_fin.sh: config 2 0.059 -> 0.028 ms against 4096 blocks, config 3
        // 0.235 -> 0.207 ms), 512 when the bucket vector is long (dual kits: 0.052 against 0.072 ms)
        const QOptVal fb_env = qopt_get(QO_FIN_BLOCKS);                          // (A/B runs)
        uint32_t blocks = (uint32_t)std::min<uint64_t>((n + 255) / 256, fb_env ? (uint64_t)std::max(1, atoi(fb_env)) : (hk.n_buckets > 2048 ? 512 : 1024));
        const bool middle = hk.scan_middle != 0;
        hipLaunchKernelGGL(k_finalize, dim3(blocks), dim3(256), fin_lds_bytes(hk.n_buckets, !middle), c->stream,
                           kp, c->recs, b->offsets, b->true_len, n, c->results, middle ? nullptr : c->counts,
                           (slim && !adapter_only) ? c->packed.bcres : nullptr, (slim && !adapter_only) ? c->packed.fin : nullptr);
        mark(c, "k_finalize");
        if (middle) {
            const uint8_t* only = nullptr;
            if (packed_kit && middle_packed_ok(hk)) {              // (also after the tiny kernels: the interiors are no handful of cells)
                if ((rc = middle_packed(c, kp, hk, b, n))) return rc;
                only = c->mid_generic;                  // interiors beyond the packed path's length classes
                mark(c, "k_middle_packed");
            }
            // the reads the packed interior scan left (interiors of more than 16 384 letters ...): on one wave per alignment
            // (kernels_tiny.inc: k_midw_*) up to MIDW_CAP of them per batch, the general kernel takes what is left after that
            c->midw_ran = false;
            if (only && hk.gap_open == hk.gap_extend && !opt_on(QO_NO_TINY) && !c->force_generic) {
                constexpr uint32_t MIDW_CAP = 2048;
                int maxb = 1;
                for (int t = 0; t < hk.nt; ++t) for (int s2 = 0; s2 < 2; ++s2) maxb = std::max(maxb, (int)hk.tpl[t].sets[s2].n);
                if (!c->midw_list) {
                    HIPCHK(q_malloc((void**)&c->midw_list, (MIDW_CAP + 1) * sizeof(uint32_t)));
                    HIPCHK(q_malloc((void**)&c->midw_tpl, (size_t)MIDW_CAP * 2 * MAX_T * 2 * sizeof(int32_t)));
                    HIPCHK(q_malloc((void**)&c->midw_recs, (size_t)MIDW_CAP * 2 * sizeof(EndRec)));
                }
                if ((size_t)maxb > c->cap_midw_sc) {
                    (void)hipFree(c->midw_sc); c->midw_sc = nullptr; c->cap_midw_sc = 0;
                    HIPCHK(q_malloc((void**)&c->midw_sc, (size_t)MIDW_CAP * 2 * 2 * (size_t)maxb * sizeof(int16_t)));
                    c->cap_midw_sc = (size_t)maxb;
                }
                uint32_t* count = c->midw_list + MIDW_CAP;
                HIPCHK(hipMemsetAsync(count, 0, sizeof(uint32_t), c->stream));
                MidWaveArgs ma{kp, b->bases, b->offsets, n, c->results, c->mid_generic, c->midw_list, count, MIDW_CAP,
                               c->midw_tpl, c->midw_recs, c->midw_sc, (uint32_t)maxb};
                hipLaunchKernelGGL(k_midw_list, dim3((n + 255) / 256), dim3(256), 0, c->stream, ma);
                hipLaunchKernelGGL(k_midw_adapter, dim3(1024), dim3(64), 0, c->stream, ma);
                hipLaunchKernelGGL(k_midw_decide, dim3(MIDW_CAP * 2 / 64), dim3(64), 0, c->stream, ma);
                hipLaunchKernelGGL(k_midw_barcode, dim3(4096), dim3(64), 0, c->stream, ma);
                hipLaunchKernelGGL(k_midw_finish, dim3(MIDW_CAP / 64), dim3(64), 0, c->stream, ma);
                c->midw_ran = true;
            }
            hipLaunchKernelGGL(k_scan_middle, dim3((n + GEN_THREADS - 1) / GEN_THREADS), dim3(GEN_THREADS), 0, c->stream,
                               kp, b->bases, b->offsets, n, c->results, only);
            hipLaunchKernelGGL(k_count, dim3(blocks), dim3(256), 0, c->stream, kp, c->results, b->offsets, b->true_len, n, c->counts);
            mark(c, "k_scan_middle");
        }
    }
    HIPCHK(hipGetLastError());
    return 0;
}

extern "C" int qcat_scan_resident(qcat_ctx* c, const qcat_kit* kit, const qcat_batch* b) {
    return scan_resident_impl(c, const_cast<qcat_kit*>(kit), b, false, 0);
}

extern "C" int qcat_ctx_synchronize(qcat_ctx* c) {
    if (!c) return set_err(QCAT_ERR_ARG, "null context");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

extern "C" int qcat_ctx_fetch_results(qcat_ctx* c, qcat_result* out, uint32_t n_reads) {
    if (!c || !out) return set_err(QCAT_ERR_ARG, "null argument");
    if (n_reads != c->last_n_reads) return set_err(QCAT_ERR_ARG, "n_reads does not match the last scan");
    HIPCHK(hipSetDevice(c->device));
    if (n_reads) HIPCHK(hipMemcpyAsync(out, c->results, (size_t)n_reads * sizeof(qcat_result), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

extern "C" int qcat_ctx_fetch_counts(qcat_ctx* c, int64_t* counts, int32_t n_buckets) {
    if (!c || !counts) return set_err(QCAT_ERR_ARG, "null argument");
    if (n_buckets != c->last_buckets) return set_err(QCAT_ERR_ARG, "bucket count does not match the last scan");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipMemcpyAsync(counts, c->counts, (size_t)n_buckets * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

// records and counts of the last scan back to a host-buffer caller: through the context's pinned block while that stays small
// (a call of the reference's own shape: 4000 reads = 96 KB) -- one DMA each and ONE wait -- else straight to the caller's arrays
static int fetch_back(qcat_ctx* c, qcat_result* out, uint32_t n_reads, int64_t* counts_add, int32_t n_buckets) {
    const size_t bytes_rec = (size_t)n_reads * sizeof(qcat_result), off_cnt = (bytes_rec + 63) / 64 * 64;
    const size_t need = off_cnt + (counts_add ? (size_t)n_buckets * 8 : 0);
    if (need > (8u << 20)) {
        int rc = qcat_ctx_fetch_results(c, out, n_reads);
        if (!rc && counts_add) {
            std::vector<int64_t> tmp((size_t)n_buckets);
            rc = qcat_ctx_fetch_counts(c, tmp.data(), n_buckets);
            if (!rc) for (size_t i = 0; i < tmp.size(); ++i) counts_add[i] += tmp[i];
        }
        return rc;
    }
    if (n_reads != c->last_n_reads || (counts_add && n_buckets != c->last_buckets)) return set_err(QCAT_ERR_ARG, "sizes do not match the last scan");
    HIPCHK(hipSetDevice(c->device));
    if (need > c->cap_pin_ret) {
        HIPCHK(hipStreamSynchronize(c->stream));
        if (c->pin_ret) (void)hipHostFree(c->pin_ret);
        c->pin_ret = nullptr; c->cap_pin_ret = 0;
        HIPCHK(hipHostMalloc((void**)&c->pin_ret, need + need / 4 + 4096));
        c->cap_pin_ret = need + need / 4 + 4096;
    }
    if (n_reads) HIPCHK_DRAIN(c->stream, hipMemcpyAsync(c->pin_ret, c->results, bytes_rec, hipMemcpyDeviceToHost, c->stream));
    if (counts_add) HIPCHK_DRAIN(c->stream, hipMemcpyAsync(c->pin_ret + off_cnt, c->counts, (size_t)n_buckets * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (n_reads) memcpy(out, c->pin_ret, bytes_rec);
    if (counts_add) {
        const int64_t* tmp = reinterpret_cast<const int64_t*>(c->pin_ret + off_cnt);
        for (int32_t i = 0; i < n_buckets; ++i) counts_add[i] += tmp[i];
    }
    return 0;
}

extern "C" void* qcat_ctx_stream(qcat_ctx* c) { return c ? (void*)c->stream : nullptr; }
extern "C" int64_t qcat_ctx_tiny_ends(const qcat_ctx* c) { return c ? (int64_t)c->last_tiny_ends : -1; }
// reads whose interior the latest --detect-middle scan put on the one-wave kernels (k_midw_*: interiors the packed interior scan
// does not take); synchronises the stream; -1: null context, 0: none / the path did not run
extern "C" int64_t qcat_ctx_middle_wave_reads(qcat_ctx* c) {
    if (!c) return -1;
    if (!c->midw_ran || !c->midw_list) return 0;
    if (hipSetDevice(c->device) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) return -1;
    uint32_t cnt = 0;
    if (hipMemcpy(&cnt, c->midw_list + 2048, sizeof cnt, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return (int64_t)std::min<uint32_t>(cnt, 2048u);
}
extern "C" int64_t qcat_ctx_graph_replays(const qcat_ctx* c) { return c ? (int64_t)(c->api_graph.replays + c->scan_graph.replays) : -1; }
// diagnostics of the latest scan of the read ends: super-tiles (2048 barcode alignments each) its bit-sliced barcode kernels took, per
// hot class summed over the (template, set) groups -- out[0] regions a few bases short of nominal (front-padded units), out[1]
// nominal regions, out[2] full windows; all 0 when the path was not taken
extern "C" int qcat_ctx_barcode_bitslice_tiles(qcat_ctx* c, uint32_t* out) {
    if (!c || !out) return set_err(QCAT_ERR_ARG, "null argument");
    out[0] = out[1] = out[2] = 0;
    if (!c->packed.bsplan || !c->packed.bs_last) return 0;
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    BsPlan hp;
    HIPCHK(hipMemcpy(&hp, c->packed.bsplan, sizeof hp, hipMemcpyDeviceToHost));
    for (int b = 0; b < JOB_BINS; ++b) {
        const int cls = b % JOB_CLASSES - (JOB_CLASSES - JOB_HOT);
        if (cls >= 0) out[cls] += hp.count[b];
    }
    return 0;
}
// diagnostics of the latest --detect-middle scan: out[0] = big tiles (2048 interiors) its bit-sliced adapter scan walked, out[1] = big
// tiles in all, out[2] = tiles of 128 interiors left to the binary16 kernel, out[3] = tiles of 128 in all (all 0: path not taken)
extern "C" int qcat_ctx_middle_bitslice_tiles(qcat_ctx* c, uint32_t* out) {
    if (!c || !out) return set_err(QCAT_ERR_ARG, "null argument");
    out[0] = out[1] = out[2] = out[3] = 0;
    if (!c->absm_last_big) return 0;
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    std::vector<int32_t> rows(c->absm_last_big);
    std::vector<uint8_t> need(c->absm_last_128);
    HIPCHK(hipMemcpy(rows.data(), c->absm_tiles, rows.size() * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(need.data(), c->absm_need, need.size(), hipMemcpyDeviceToHost));
    for (int32_t r : rows) if (r > 0) ++out[0];
    out[1] = c->absm_last_big;
    for (uint8_t f : need) if (f) ++out[2];
    out[3] = c->absm_last_128;
    return 0;
}
extern "C" void* qcat_ctx_counts_devptr(qcat_ctx* c) { return c ? c->counts : nullptr; }
extern "C" void* qcat_ctx_results_devptr(qcat_ctx* c) { return c ? c->results : nullptr; }

extern "C" int qcat_ctx_last_timing(qcat_ctx* c, const char** names, float* ms, int cap) {
    if (!c) return set_err(QCAT_ERR_ARG, "null context");
    if (!c->timing) return 0;
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    // average per mark over the recorded scans that have the same mark list as the newest one
    const int last = c->ring_used - 1;
    if (last < 0) return 0;
    const int n = std::min(cap, c->ring_marks[last]);
    for (int i = 0; i < n; ++i) { names[i] = c->timed_name[last][i]; ms[i] = 0.f; }
    int scans = 0;
    for (int r = 0; r <= last; ++r) {
        if (c->ring_marks[r] != c->ring_marks[last]) continue;
        bool same = true;
        for (int i = 0; i < n; ++i) same = same && c->timed_name[r][i] == c->timed_name[last][i];
        if (!same) continue;
        for (int i = 0; i < n; ++i) { float t = 0.f; HIPCHK(hipEventElapsedTime(&t, c->evr[r][i], c->evr[r][i + 1])); ms[i] += t; }
        ++scans;
    }
    for (int i = 0; i < n; ++i) ms[i] /= (float)scans;
    c->ring_used = 0; c->n_timed = 0; c->ev = nullptr;
    return n;
}

// ------------------------------------------------------------------------------------------
// batches: upload / download / synthetic
// ------------------------------------------------------------------------------------------
extern "C" void qcat_batch_destroy(qcat_batch* b) {
    if (!b) return;
    int cur = 0; (void)hipGetDevice(&cur);
    (void)hipSetDevice(b->device);
    if (!b->borrowed) { (void)hipFree(b->bases_alloc); (void)hipFree(b->offsets); (void)hipFree(b->true_len); }
    (void)hipSetDevice(cur);
    delete b;
}

extern "C" int qcat_batch_info(const qcat_batch* b, uint32_t* n_reads, uint64_t* n_bases) {
    if (!b) return set_err(QCAT_ERR_ARG, "null batch");
    if (n_reads) *n_reads = b->n_reads;
    if (n_bases) *n_bases = b->n_bases;
    return 0;
}

extern "C" int qcat_batch_upload(qcat_ctx* c, const uint8_t* bases, const uint64_t* offsets,
                                 uint32_t n_reads, qcat_batch** out) {
    if (!c || !offsets || !out || (!bases && n_reads && offsets[n_reads] > 0))
        return set_err(QCAT_ERR_ARG, "qcat_batch_upload: null argument");
    if (offsets[0] != 0) return set_err(QCAT_ERR_ARG, "offsets[0] must be 0");
    for (uint32_t r = 0; r < n_reads; ++r)
        if (offsets[r + 1] < offsets[r]) return set_err(QCAT_ERR_ARG, "offsets must be non-decreasing");
    HIPCHK(hipSetDevice(c->device));
    qcat_batch* b = new qcat_batch();
    BatchGuard guard(b);
    b->device = c->device; b->n_reads = n_reads; b->n_bases = offsets[n_reads];
    hipError_t e1 = hipMalloc((void**)&b->bases_alloc, b->n_bases + 2 * BATCH_SLACK);
    if (e1 == hipSuccess) b->bases = b->bases_alloc + BATCH_SLACK;
    hipError_t e2 = hipMalloc((void**)&b->offsets, ((size_t)n_reads + 1) * 8);
    if (e1 != hipSuccess || e2 != hipSuccess) return set_err(QCAT_ERR_NOMEM, "hipMalloc failed for batch");
    if (b->n_bases) HIPCHK(hipMemcpyAsync(b->bases, bases, b->n_bases, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(b->offsets, offsets, ((size_t)n_reads + 1) * 8, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    *out = guard.release();
    return 0;
}

// Host-buffer entry points scan only the first / last max_align_length bases of a read, so that is
// all they send over PCIe: every read is compacted to head + tail (reads up to 2n stay whole, which
// keeps the two windows byte-identical), the real length travels in a separate array for the trims.
// The compaction runs on a few host threads into pinned memory.  --detect-middle needs whole reads.
// `ptrs` / `lens` (both or neither): the reads as one pointer and one length per read instead of the concatenated form
// (qcat_scan_batch_ptrs: a host language that holds a string object per read passes the objects' own buffers)
static int batch_upload_windows(qcat_ctx* c, const qcat_kit* kit, const uint8_t* bases, const uint64_t* offsets,
                                uint32_t n_reads, qcat_batch** out, const uint8_t* const* ptrs = nullptr, const uint64_t* lens = nullptr) {
    const DevKit& hk = kit->hk.dk;
    // the paths that take WHOLE reads (--detect-middle: the interior scan; QCAT_HIP_FULL_UPLOAD): round 6 -- through the same pinned
    // staging, host threads and context-owned device buffers as the windows (a read is "compacted" to all of itself).  Before,
    // they were concatenated by one thread into pageable memory and went through qcat_batch_upload: a hipMalloc / hipFree and a
    // pageable copy per call -- 21 ms per 173 000-read segment of the driver's file loop, whose scan takes 1 ms.  Batches
    // of more than a gigabyte keep the plain upload (the staging is pinned memory and stays with the context).
    const bool whole = hk.scan_middle || opt_on(QO_FULL_UPLOAD);
    if (whole && c && (ptrs || offsets)) {
        uint64_t tot = 0;
        if (ptrs) for (uint32_t r = 0; r < n_reads; ++r) tot += lens[r];
        else tot = offsets[n_reads];
        if (tot > (1ull << 30)) {
            std::vector<uint8_t> cat;
            std::vector<uint64_t> cat_off;
            if (ptrs) {
                cat_off.resize((size_t)n_reads + 1);
                tot = 0;
                for (uint32_t r = 0; r < n_reads; ++r) {
                    if (lens[r] && !ptrs[r]) return set_err(QCAT_ERR_ARG, "null read pointer");
                    if (lens[r] > 0xFFFFFFFFull) return set_err(QCAT_ERR_UNSUPPORTED, "read longer than 4 Gb");
                    cat_off[r] = tot; tot += lens[r];
                }
                cat_off[n_reads] = tot;
                cat.resize((size_t)tot + 1);
                for (uint32_t r = 0; r < n_reads; ++r) if (lens[r]) memcpy(cat.data() + cat_off[r], ptrs[r], (size_t)lens[r]);
                bases = cat.data(); offsets = cat_off.data();
            }
            return qcat_batch_upload(c, bases, offsets, n_reads, out);
        }
    }
    if (!c || !out || (!ptrs && (!offsets || (!bases && offsets[n_reads] > 0)))) return set_err(QCAT_ERR_ARG, "qcat_scan_batch: null argument");
    if (!ptrs && offsets[0] != 0) return set_err(QCAT_ERR_ARG, "offsets[0] must be 0");
    HIPCHK(hipSetDevice(c->device));
    const uint64_t n = (uint64_t)hk.max_align;
    const bool both = hk.ends == QCAT_ENDS_BOTH;
    const uint64_t keep = whole ? ~0ull : (both ? 2 * n : n);
    if ((size_t)n_reads + 1 > c->cap_pin_reads) {
        if (c->pin_offsets) (void)hipHostFree(c->pin_offsets);
        if (c->pin_len) (void)hipHostFree(c->pin_len);
        c->pin_offsets = nullptr; c->pin_len = nullptr; c->cap_pin_reads = 0;
        HIPCHK(hipHostMalloc((void**)&c->pin_offsets, ((size_t)n_reads + 1) * 8));
        HIPCHK(hipHostMalloc((void**)&c->pin_len, ((size_t)n_reads + 1) * 4));
        c->cap_pin_reads = (size_t)n_reads + 1;
        g_alloc_gen.fetch_add(1);                                      // (kernels of a handful-of-reads call read the staging in place)
    }
    uint64_t total = 0;
    c->pin_offsets[0] = 0;
    for (uint32_t r = 0; r < n_reads; ++r) {
        if (!ptrs && offsets[r + 1] < offsets[r]) return set_err(QCAT_ERR_ARG, "offsets must be non-decreasing");
        const uint64_t len = ptrs ? lens[r] : offsets[r + 1] - offsets[r];
        if (ptrs && len && !ptrs[r]) return set_err(QCAT_ERR_ARG, "null read pointer");
        if (len > 0xFFFFFFFFull) return set_err(QCAT_ERR_UNSUPPORTED, "read longer than 4 Gb");
        c->pin_len[r] = (uint32_t)len;
        total += len <= keep ? len : keep;
        c->pin_offsets[r + 1] = total;
    }
    if (total + 1 > c->cap_pin_bases) {
        if (c->pin_bases) (void)hipHostFree(c->pin_bases);
        c->pin_bases = nullptr; c->cap_pin_bases = 0;
        HIPCHK(hipHostMalloc((void**)&c->pin_bases, total + 1 + 2 * BATCH_SLACK));      // (slack either side like a device batch)
        memset(c->pin_bases, 0, BATCH_SLACK);
        c->cap_pin_bases = total + 1;
        g_alloc_gen.fetch_add(1);
    }
    uint8_t* const pin_data = c->pin_bases + BATCH_SLACK;
    // the context's own device staging (the batch is a view of it: one host-buffer call at a time per context)
    if (total + 2 * BATCH_SLACK > c->cap_hb_bases) {
        (void)hipFree(c->hb_bases); c->hb_bases = nullptr; c->cap_hb_bases = 0;
        const size_t want = (total + 2 * BATCH_SLACK) * 5 / 4;
        if (q_malloc((void**)&c->hb_bases, want) != hipSuccess) return set_err(QCAT_ERR_NOMEM, "hipMalloc failed for batch");
        c->cap_hb_bases = want;
    }
    if ((size_t)n_reads + 1 > c->cap_hb_reads) {
        (void)hipFree(c->hb_offsets); (void)hipFree(c->hb_len); c->hb_offsets = nullptr; c->hb_len = nullptr; c->cap_hb_reads = 0;
        const size_t want = ((size_t)n_reads + 1) * 5 / 4;
        if (q_malloc((void**)&c->hb_offsets, want * 8) != hipSuccess || q_malloc((void**)&c->hb_len, want * 4) != hipSuccess)
            return set_err(QCAT_ERR_NOMEM, "hipMalloc failed for batch");
        c->cap_hb_reads = want;
    }
    uint8_t* const dev_data = c->hb_bases + BATCH_SLACK;
    bool sent = false;                                   // the bases have gone to the device slice by slice
    {
        const unsigned nthreads = std::min<unsigned>(host_threads(), (unsigned)std::max<uint64_t>(1u, std::max<uint64_t>(n_reads / 16384u, total >> 23)));
        auto work = [&](uint32_t r0, uint32_t r1) {
            for (uint32_t r = r0; r < r1; ++r) {
                const uint8_t* src = ptrs ? ptrs[r] : bases + offsets[r];
                const uint64_t len = ptrs ? lens[r] : offsets[r + 1] - offsets[r];
                uint8_t* dst = pin_data + c->pin_offsets[r];
                if (len <= keep) { memcpy(dst, src, len); continue; }
                memcpy(dst, src, n);
                if (both) memcpy(dst + n, src + len - n, n);
            }
        };
        std::vector<std::thread> pool;
        // a big staging (round 6: whole reads of --detect-middle are 130 MB per segment of the file loop) in slices of ~8 MB that
        // the workers draw from a counter: this thread sends every finished run of slices on while the others are still being
        // copied -- the transfer hides behind the compaction instead of following it
        const uint64_t slice_bytes = 8ull << 20;
        if (nthreads >= 2 && total >= 4 * slice_bytes) {
            std::vector<uint32_t> bound;                 // slice j = reads [bound[j], bound[j + 1])
            bound.push_back(0);
            for (uint64_t at = slice_bytes; at < total; at += slice_bytes) {
                const uint32_t r = (uint32_t)(std::lower_bound(c->pin_offsets, c->pin_offsets + n_reads, at) - c->pin_offsets);
                if (r > bound.back() && r < n_reads) bound.push_back(r);
            }
            bound.push_back(n_reads);
            const size_t n_slices = bound.size() - 1;
            std::unique_ptr<std::atomic<uint8_t>[]> done(new std::atomic<uint8_t>[n_slices]);
            for (size_t j = 0; j < n_slices; ++j) done[j].store(0, std::memory_order_relaxed);
            std::atomic<size_t> next{0};
            auto drain = [&] {
                for (;;) {
                    const size_t j = next.fetch_add(1);
                    if (j >= n_slices) break;
                    work(bound[j], bound[j + 1]);
                    done[j].store(1, std::memory_order_release);
                }
            };
            for (unsigned t = 0; t < nthreads; ++t) {
                try { pool.emplace_back(drain); }
                catch (const std::exception&) { break; }
            }
            if (pool.empty()) drain();
            hipError_t ce = hipSuccess;
            for (size_t j = 0; j < n_slices && ce == hipSuccess; ) {
                while (!done[j].load(std::memory_order_acquire)) std::this_thread::yield();
                size_t k = j + 1;
                while (k < n_slices && done[k].load(std::memory_order_acquire)) ++k;
                const uint64_t o0 = c->pin_offsets[bound[j]], o1 = c->pin_offsets[bound[k]];
                if (o1 > o0) ce = hipMemcpyAsync(dev_data + o0, pin_data + o0, o1 - o0, hipMemcpyHostToDevice, c->stream);
                j = k;
            }
            for (auto& th : pool) th.join();
            if (ce != hipSuccess) { (void)hipStreamSynchronize(c->stream); return set_err(QCAT_ERR_DEVICE, std::string("upload: ") + hipGetErrorString(ce)); }
            sent = true;
        } else {
            const uint32_t per = (n_reads + nthreads - 1) / nthreads;
            uint32_t done_to = std::min<uint32_t>(n_reads, per);          // [0, per) is this thread's share
            for (unsigned t = 1; t < nthreads; ++t) {
                const uint32_t r0 = std::min<uint32_t>(n_reads, t * per), r1 = std::min<uint32_t>(n_reads, r0 + per);
                if (r0 >= r1) break;
                try { pool.emplace_back(work, r0, r1); }
                catch (const std::exception&) { break; }                 // no more threads: the rest runs here
                done_to = r1;
            }
            work(0, std::min<uint32_t>(n_reads, per));
            if (done_to < n_reads) work(done_to, n_reads);
            for (auto& th : pool) th.join();
        }
    }
    qcat_batch* b = new qcat_batch();
    BatchGuard guard(b);
    b->device = c->device; b->n_reads = n_reads; b->n_bases = total; b->borrowed = true;
    // a handful of reads (detect_barcode on one read): the kernels read the pinned staging in place -- host memory the device
    // maps at the same address -- instead of waiting for three copies of a few hundred bytes (~4.5 us each in the call's chain)
    if (n_reads <= 64 && !whole && !sent && !opt_on(QO_NO_ZERO_COPY)) {
        memset(pin_data + total, 0, 1 + BATCH_SLACK);
        b->bases_alloc = c->pin_bases; b->bases = pin_data; b->offsets = c->pin_offsets; b->true_len = c->pin_len;
        *out = guard.release();
        return 0;
    }
    b->bases_alloc = c->hb_bases; b->bases = dev_data; b->offsets = c->hb_offsets; b->true_len = c->hb_len;
    if (total && !sent) HIPCHK(hipMemcpyAsync(b->bases, pin_data, total, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(b->offsets, c->pin_offsets, ((size_t)n_reads + 1) * 8, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(b->true_len, c->pin_len, (size_t)n_reads * 4, hipMemcpyHostToDevice, c->stream));
    // (no synchronisation here: every caller hands results back to the host and drains the stream for that before it returns,
    //  so the pinned staging is free again when the next call fills it -- one round trip less per 4000-read batch)
    *out = guard.release();
    return 0;
}

extern "C" int qcat_batch_download(qcat_ctx* c, const qcat_batch* b, uint8_t* bases, uint64_t* offsets) {
    if (!c || !b) return set_err(QCAT_ERR_ARG, "null argument");
    HIPCHK(hipSetDevice(c->device));
    if (offsets) HIPCHK(hipMemcpyAsync(offsets, b->offsets, ((size_t)b->n_reads + 1) * 8, hipMemcpyDeviceToHost, c->stream));
    if (bases && b->n_bases) HIPCHK(hipMemcpyAsync(bases, b->bases, b->n_bases, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

// --- synthetic reads ------------------------------------------------------------------------
struct SynthSetup {
    qsynth::Params p;
    qsynth::Tpl t5, t3;
    bool has5 = false, has3 = false;
};

static int synth_setup(const HostKit& hk, const qcat_synth_params* sp, const char* ascii_base, SynthSetup* s, std::string* err) {
    if (!sp) { *err = "null synth params"; return QCAT_ERR_ARG; }
    if (sp->lead_max < sp->lead_min) { *err = "lead_max < lead_min"; return QCAT_ERR_ARG; }
    if (sp->error_rate < 0.f || sp->error_rate > 1.f || sp->no_adapter_fraction < 0.f || sp->no_adapter_fraction > 1.f) {
        *err = "rates must be in [0,1]"; return QCAT_ERR_ARG;
    }
    s->p.seed = sp->seed; s->p.insert_len = sp->insert_len; s->p.lead_min = sp->lead_min; s->p.lead_max = sp->lead_max;
    s->p.thr_err = (uint32_t)(sp->error_rate * 16777216.0f);
    s->p.thr_none = (uint32_t)(sp->no_adapter_fraction * 16777216.0f);
    auto fill = [&](int t, qsynth::Tpl* o) -> bool {
        if (t < 0) return false;
        const DevTpl& p = hk.dk.tpl[t];
        o->seq = ascii_base + hk.ascii_tpl_off[t]; o->len = p.len;
        for (int i = 0; i < 2; ++i) {
            o->bc_start[i] = hk.bc_start[t][i]; o->bc_len[i] = p.sets[i].n > 0 ? p.bc_len[i] : 0;
            o->n[i] = p.sets[i].n;
            o->sets[i] = p.sets[i].n > 0 ? ascii_base + hk.ascii_set_off[t][i] : nullptr;
        }
        return true;
    };
    if (sp->tpl_5p >= hk.dk.nt || sp->tpl_3p >= hk.dk.nt) { *err = "synth template index out of range"; return QCAT_ERR_ARG; }
    s->has5 = fill(sp->tpl_5p, &s->t5);
    s->has3 = fill(sp->tpl_3p, &s->t3);
    return 0;
}

extern "C" int64_t qcat_synth_read(const qcat_kit* kit, const qcat_synth_params* sp, uint64_t index,
                                   uint8_t* buf, uint64_t cap) {
    if (!kit) return set_err(QCAT_ERR_ARG, "null kit");
    SynthSetup s; std::string err;
    int rc = synth_setup(kit->hk, sp, kit->hk.ascii.data(), &s, &err);
    if (rc) return set_err(rc, err);
    qsynth::StoreSink sink(buf, buf ? cap : 0);
    qsynth::generate(s.p, index, s.has5 ? &s.t5 : nullptr, s.has3 ? &s.t3 : nullptr, sink);
    return (int64_t)sink.n;
}

__global__ void k_synth_lengths(qsynth::Params p, qsynth::Tpl t5, qsynth::Tpl t3, int has5, int has3,
                                uint64_t first, uint32_t n, uint64_t* lens) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    qsynth::CountSink sink;
    qsynth::generate(p, first + i, has5 ? &t5 : nullptr, has3 ? &t3 : nullptr, sink);
    lens[i] = sink.n;
}

__global__ void k_synth_write(qsynth::Params p, qsynth::Tpl t5, qsynth::Tpl t3, int has5, int has3,
                              uint64_t first, uint32_t n, const uint64_t* offsets, uint8_t* bases) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    qsynth::StoreSink sink(bases + offsets[i], offsets[i + 1] - offsets[i]);
    qsynth::generate(p, first + i, has5 ? &t5 : nullptr, has3 ? &t3 : nullptr, sink);
}

extern "C" int qcat_batch_synthesize(qcat_ctx* c, const qcat_kit* ckit, const qcat_synth_params* sp, qcat_batch** out) {
    if (!c || !ckit || !sp || !out) return set_err(QCAT_ERR_ARG, "null argument");
    qcat_kit* kit = const_cast<qcat_kit*>(ckit);
    HIPCHK(hipSetDevice(c->device));
    KitOnDevice* kd = nullptr;
    int rc = kit_on_device(kit, c->device, &kd);
    if (rc) return rc;
    SynthSetup s; std::string err;
    if ((rc = synth_setup(kit->hk, sp, kd->ascii, &s, &err))) return set_err(rc, err);
    const uint32_t n = sp->n_reads;
    DevTemp lens_buf;
    HIPCHK(lens_buf.alloc(((size_t)n + 1) * 8));
    uint64_t* d_lens = lens_buf.as<uint64_t>();
    uint32_t blocks = (n + 255) / 256;
    // the read index space is global: `seed` identifies the data set, reads [first, first+n) are
    // produced here (first = 0; shards pass distinct seeds or use qcat_synth_read for offsets)
    if (n) hipLaunchKernelGGL(k_synth_lengths, dim3(blocks), dim3(256), 0, c->stream, s.p, s.t5, s.t3,
                              (int)s.has5, (int)s.has3, (uint64_t)0, n, d_lens);
    std::vector<uint64_t> lens((size_t)n + 1, 0), offs((size_t)n + 1, 0);
    if (n) HIPCHK(hipMemcpyAsync(lens.data(), d_lens, (size_t)n * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    for (uint32_t i = 0; i < n; ++i) offs[i + 1] = offs[i] + lens[i];
    qcat_batch* b = new qcat_batch();
    BatchGuard guard(b);
    b->device = c->device; b->n_reads = n; b->n_bases = offs[n];
    hipError_t e1 = hipMalloc((void**)&b->bases_alloc, b->n_bases + 2 * BATCH_SLACK);
    if (e1 == hipSuccess) b->bases = b->bases_alloc + BATCH_SLACK;
    hipError_t e2 = hipMalloc((void**)&b->offsets, ((size_t)n + 1) * 8);
    if (e1 != hipSuccess || e2 != hipSuccess) return set_err(QCAT_ERR_NOMEM, "hipMalloc failed for synthetic batch");
    HIPCHK(hipMemcpyAsync(b->offsets, offs.data(), ((size_t)n + 1) * 8, hipMemcpyHostToDevice, c->stream));
    if (n) hipLaunchKernelGGL(k_synth_write, dim3(blocks), dim3(256), 0, c->stream, s.p, s.t5, s.t3,
                              (int)s.has5, (int)s.has3, (uint64_t)0, n, b->offsets, b->bases);
    hipError_t es = hipStreamSynchronize(c->stream);
    if (es == hipSuccess) es = hipGetLastError();
    if (es != hipSuccess) return set_err(QCAT_ERR_DEVICE, std::string("qcat_batch_synthesize: ") + hipGetErrorString(es));
    *out = guard.release();
    return 0;
}

// ------------------------------------------------------------------------------------------
// host-buffer entry points
// ------------------------------------------------------------------------------------------
static void fill_trace(const HostKit& hk, const EndRec& r, const int32_t* tplraw, const int32_t* tplend, qcat_end_trace* t) {
    memset(t, 0, sizeof *t);
    t->window_len = r.window_len;
    for (int i = 0; i < MAX_T; ++i) { t->tpl_raw[i] = tplraw[i]; t->tpl_end[i] = tplend[i]; }
    t->best_tpl = r.best_tpl; t->best_end = r.best_end; t->best_raw = r.best_raw; t->used_tpl = r.used_tpl;
    t->region_path = r.region_path;
    for (int s = 0; s < 2; ++s) {
        t->region_start[s] = r.region_start[s]; t->region_len[s] = r.region_len[s];
        t->bc_idx[s] = r.bc_idx[s]; t->bc_raw[s] = r.bc_raw[s];
    }
    const DevTpl& p = hk.dk.tpl[r.used_tpl];
    if (hk.dk.mode == QCAT_MODE_EPI2ME) {
        int ae = r.best_end + p.trim_offset;
        t->adapter_end = ae > r.window_len ? r.window_len : ae;
    } else if (hk.dk.mode == QCAT_MODE_SIMPLE) {
        t->adapter_end = (r.bc_idx[0] >= 0 && r.bc_raw[0] >= p.sets[0].min_raw_pass) ? r.best_end : 0;
    } else {
        t->adapter_end = (r.bc_idx[0] >= 0 && r.bc_idx[1] >= 0) ? r.best_end : 0;
    }
}

// qcat_scan_batch over host buffers as a chunked pipeline (host_pipeline.inc).  Returns 1 when the batch
// does not qualify (small, --detect-middle, disabled) and the caller should take the one-shot path.
// where the reads of a host batch lie: concatenated (bases + offsets, the form of qcat_scan_batch) or scattered through a
// mapped FASTQ file (one FqRec per read, fastq_host.inc) -- the pipeline only ever reads head and tail of a read in place
namespace qk { struct FqRec { uint64_t title, seq, qual; uint32_t title_len, seq_len; }; }
struct ReadView {
    const uint8_t* bases;
    const uint64_t* offsets;           // concatenated form (recs == nullptr)
    const qk::FqRec* recs;             // FASTQ form: bases = the mapping, read r = recs[r]
    inline uint64_t start(uint32_t r) const { return recs ? recs[r].seq : offsets[r]; }
    inline uint64_t len(uint32_t r) const { return recs ? (uint64_t)recs[r].seq_len : offsets[r + 1] - offsets[r]; }
};

static int scan_batch_pipelined(qcat_ctx* c, qcat_kit* kit, const ReadView& rv,
                                uint32_t n_reads, qcat_result* out, int64_t* counts) {
    const DevKit& hk = kit->hk.dk;
    const uint8_t* bases = rv.bases;
    if (hk.scan_middle || n_reads < 32768 || opt_on(QO_FULL_UPLOAD) || opt_on(QO_NO_PIPELINE)) return 1;
    if (!rv.recs) {
        if (!bases && rv.offsets[n_reads] > 0) return set_err(QCAT_ERR_ARG, "qcat_scan_batch: null argument");
        if (rv.offsets[0] != 0) return set_err(QCAT_ERR_ARG, "offsets[0] must be 0");
    }
    HIPCHK(hipSetDevice(c->device));
    const uint64_t n = (uint64_t)hk.max_align;
    const bool both = hk.ends == QCAT_ENDS_BOTH;
    const uint64_t keep = both ? 2 * n : n;
    const auto t_entry = std::chrono::steady_clock::now();
    if (!c->pipe) {
        c->pipe = new HostPipeline();
        c->pipe->pool = new HostPool(host_threads() - 1);
        // the copy stream gets the highest priority: the runtime multiplexes streams of one priority onto a few
        // hardware queues, and a copy queued behind a 20 ms persistent barcode kernel of a side stream would
        // not start before that kernel ends (measured: no copy/compute overlap at all on a default stream)
        int prio_low = 0, prio_high = 0;
        HIPCHK(hipDeviceGetStreamPriorityRange(&prio_low, &prio_high));
        HIPCHK(hipStreamCreateWithPriority(&c->pipe->copy, hipStreamNonBlocking, prio_high));
        for (PipeStage& st : c->pipe->st) {
            HIPCHK(hipEventCreateWithFlags(&st.copied, hipEventDisableTiming));
            HIPCHK(hipEventCreateWithFlags(&st.scanned, hipEventDisableTiming));
        }
    }
    HostPipeline* p = c->pipe;
    // chunks of 256 k .. 1 M reads (a quarter of the batch): big enough to fill the chip (1 000+ tiles of 128
    // alignments), small enough that the first chunk's compaction and the last chunk's scan -- the only
    // stages nothing overlaps -- are a small part of the call
    const QOptVal ce = qopt_get(QO_PIPELINE_CHUNK);
    // (heavier per-chunk launches lose less to the tails of the persistent barcode kernels: large batches take 1 M-read chunks)
    // (a chunk costs ~0.3 ms of host time in launches and copies whatever its size: 1 M reads of a small kit run
    // 166 M reads/s in four chunks, 137 M in eight, 97 M in sixteen)
    // (reads of a FASTQ mapping: a one-shot process pays for every pinned byte it sets up, 0.2 ms per MiB, and the host side
    // bounds the rate anyway -- 256 k-read chunks)
    const uint32_t chunk = ce ? (uint32_t)std::max(4096, atoi(ce))
                              : (rv.recs ? 262144u : std::min<uint32_t>(1048576u, std::max<uint32_t>(262144u, n_reads / 4u)));
    // chunk boundaries: the first and the last chunk are a third of the others -- nothing overlaps the first chunk's
    // compaction and the last chunk's upload + scan + download
    std::vector<uint32_t> cuts;
    cuts.push_back(0);
    if (!ce && n_reads > 2 * chunk) {
        const uint32_t edge = chunk / 3, mid = n_reads - 2 * edge, m = (mid + chunk - 1) / chunk;
        for (uint32_t q = 0; q <= m; ++q) cuts.push_back(edge + (uint32_t)((uint64_t)mid * q / m));    // equal middle chunks
        cuts.push_back(n_reads);
    } else {
        for (uint32_t pos = chunk; pos < n_reads; pos += chunk) cuts.push_back(pos);
        cuts.push_back(n_reads);
    }
    const uint32_t n_chunks = (uint32_t)cuts.size() - 1;
    // staging for the LARGEST chunk of this call, and for the second slot only when there is a second chunk (round 6: a segment of
    // the file loop is one chunk of 40-180 k reads -- sizing both slots for 256 k reads pinned 157 MB, 0.2 ms per MiB, in front
    // of the first segment's scan: 42 ms of a 160 ms run); a later, bigger call grows them with headroom
    uint32_t largest = 0;
    for (uint32_t q = 0; q < n_chunks; ++q) largest = std::max(largest, cuts[q + 1] - cuts[q]);
    for (PipeStage& st : p->st) {
        if (&st != &p->st[0] && n_chunks < 2) break;
        size_t need_reads = (size_t)largest + 1, need_bases = (size_t)largest * keep + 2 * BATCH_SLACK;
        if (need_reads > st.cap_reads && st.cap_reads) need_reads = std::min<size_t>((size_t)chunk + 1, need_reads + need_reads / 2);
        if (need_bases > st.cap_bases && st.cap_bases) need_bases = std::min<size_t>((size_t)chunk * keep + 2 * BATCH_SLACK, need_bases + need_bases / 2);
        if (need_reads > st.cap_reads) {
            if (st.pin_offsets) (void)hipHostFree(st.pin_offsets);
            if (st.pin_len) (void)hipHostFree(st.pin_len);
            if (st.pin_results) (void)hipHostFree(st.pin_results);
            (void)hipFree(st.dev_offsets); (void)hipFree(st.dev_len);
            st.pin_offsets = nullptr; st.pin_len = nullptr; st.pin_results = nullptr; st.dev_offsets = nullptr; st.dev_len = nullptr; st.cap_reads = 0;
            HIPCHK(hipHostMalloc((void**)&st.pin_offsets, need_reads * 8));
            HIPCHK(hipHostMalloc((void**)&st.pin_len, need_reads * 4));
            HIPCHK(hipHostMalloc((void**)&st.pin_results, need_reads * sizeof(qcat_result)));
            HIPCHK(hipMalloc((void**)&st.dev_offsets, need_reads * 8));
            HIPCHK(hipMalloc((void**)&st.dev_len, need_reads * 4));
            st.cap_reads = need_reads;
        }
        if (need_bases > st.cap_bases) {
            if (st.pin_bases) (void)hipHostFree(st.pin_bases);
            (void)hipFree(st.dev_bases_alloc);
            st.pin_bases = nullptr; st.dev_bases_alloc = nullptr; st.cap_bases = 0;
            HIPCHK(hipHostMalloc((void**)&st.pin_bases, need_bases));
            HIPCHK(hipMalloc((void**)&st.dev_bases_alloc, need_bases));
            st.cap_bases = need_bases;
        }
    }
    struct TimingOff {                                   // per-kernel events describe one scan, not a chunk train
        qcat_ctx* c; bool was;
        explicit TimingOff(qcat_ctx* c_) : c(c_), was(c_->timing) { c->timing = false; }
        ~TimingOff() { c->timing = was; }
    } timing_off(c);
    int rc = 0;
    const bool trace = opt_on(QO_PIPELINE_TRACE);
    double t_wait = 0, t_prefix = 0, t_compact = 0, t_enqueue = 0;
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_begin = now();
    // a slot's records go to the caller's array once its scan is known to be done
    auto flush_results = [&](PipeStage& st) {
        if (!st.res_count) return;
        const uint32_t cnt = st.res_count, first = st.res_first;
        const int parts = (int)std::min<uint32_t>(p->pool->size(), std::max<uint32_t>(1u, cnt / 65536u));
        const uint32_t per = (cnt + parts - 1) / parts;
        p->pool->run(parts, [&](int part) {
            const uint32_t a = std::min<uint32_t>(cnt, (uint32_t)part * per), b2 = std::min<uint32_t>(cnt, a + per);
            if (b2 > a) memcpy(out + first + a, st.pin_results + a, (size_t)(b2 - a) * sizeof(qcat_result));
        });
        st.res_count = 0;
    };
    for (PipeStage& st : p->st) st.res_count = 0;
    for (uint32_t ci = 0; ci < n_chunks && !rc; ++ci) {
        PipeStage& st = p->st[ci & 1];
        double t0 = now();
        const uint32_t r0 = cuts[ci], nr = cuts[ci + 1] - r0;
        if (ci >= 2) HIPCHK(hipEventSynchronize(st.copied));          // this slot's pinned staging has been read
        t_wait += now() - t0; t0 = now();
        // offsets + real lengths of the compacted chunk: per-part sums in parallel, a serial scan over the parts,
        // then every part writes its offsets and copies its reads
        const int parts = (int)std::min<uint32_t>(std::min<uint32_t>(p->pool->size() * 4, 256u), std::max<uint32_t>(1u, nr / 4096u));
        const uint32_t per = (nr + parts - 1) / parts;
        uint64_t part_total[256];
        int part_err[256];
        p->pool->run(parts, [&](int part) {
            const uint32_t a = std::min<uint32_t>(nr, (uint32_t)part * per), b2 = std::min<uint32_t>(nr, a + per);
            uint64_t sum = 0; int err = 0;
            for (uint32_t r = a; r < b2; ++r) {
                if (!rv.recs && rv.offsets[r0 + r + 1] < rv.offsets[r0 + r]) { err = 1; break; }
                const uint64_t len = rv.len(r0 + r);
                if (len > 0xFFFFFFFFull) { err = 2; break; }
                sum += len <= keep ? len : keep;
            }
            part_total[part] = sum; part_err[part] = err;
        });
        uint64_t total = 0;
        for (int q = 0; q < parts; ++q) {
            if (part_err[q] == 1) rc = set_err(QCAT_ERR_ARG, "offsets must be non-decreasing");
            else if (part_err[q] == 2) rc = set_err(QCAT_ERR_UNSUPPORTED, "read longer than 4 Gb");
            const uint64_t t = part_total[q]; part_total[q] = total; total += t;
        }
        if (rc) break;
        st.pin_offsets[0] = 0;
        t_prefix += now() - t0; t0 = now();
        p->pool->run(parts, [&](int part) {
            const uint32_t a = std::min<uint32_t>(nr, (uint32_t)part * per), b2 = std::min<uint32_t>(nr, a + per);
            uint64_t pos = part_total[part];
            constexpr uint32_t AHEAD = 12;                           // reads: the windows are 150 B out of every ~700, each a cache
            for (uint32_t r = a; r < b2; ++r) {                      // miss the hardware prefetcher does not see coming
                if (r + AHEAD < b2) {
                    const uint64_t px = rv.start(r0 + r + AHEAD), py = px + rv.len(r0 + r + AHEAD);
                    const uint8_t* ps = bases + px;
                    __builtin_prefetch(ps, 0, 0); __builtin_prefetch(ps + 64, 0, 0); __builtin_prefetch(ps + 128, 0, 0);
                    if (both && py - px > keep) {
                        const uint8_t* pt = bases + py - n;
                        __builtin_prefetch(pt, 0, 0); __builtin_prefetch(pt + 64, 0, 0); __builtin_prefetch(pt + 128, 0, 0);
                    }
                }
                const uint64_t x = rv.start(r0 + r);
                const uint64_t len = rv.len(r0 + r);
                const uint8_t* src = bases + x;
                uint8_t* dst = st.pin_bases + pos;
                st.pin_len[r] = (uint32_t)len;
                if (len <= keep) { memcpy(dst, src, len); pos += len; }
                else { memcpy(dst, src, n); if (both) memcpy(dst + n, src + len - n, n); pos += keep; }
                st.pin_offsets[r + 1] = pos;
            }
        });
        t_compact += now() - t0; t0 = now();
        // this slot's device buffers are free again once the scan of chunk ci - 2 is done: waited for on the HOST, so
        // that the copy stream never holds a barrier packet behind which its copies would queue up
        if (ci >= 2) { HIPCHK(hipEventSynchronize(st.scanned)); flush_results(st); }
        if (total) HIPCHK(hipMemcpyAsync(st.dev_bases_alloc + BATCH_SLACK, st.pin_bases, total, hipMemcpyHostToDevice, p->copy));
        HIPCHK(hipMemcpyAsync(st.dev_offsets, st.pin_offsets, ((size_t)nr + 1) * 8, hipMemcpyHostToDevice, p->copy));
        HIPCHK(hipMemcpyAsync(st.dev_len, st.pin_len, (size_t)nr * 4, hipMemcpyHostToDevice, p->copy));
        HIPCHK(hipEventRecord(st.copied, p->copy));
        HIPCHK(hipStreamWaitEvent(c->stream, st.copied, 0));
        qcat_batch view;                                               // the chunk as a resident batch (no ownership)
        view.device = c->device; view.n_reads = nr; view.n_bases = total;
        view.bases_alloc = st.dev_bases_alloc; view.bases = st.dev_bases_alloc + BATCH_SLACK;
        view.offsets = st.dev_offsets; view.true_len = st.dev_len;
        rc = scan_resident_impl(c, kit, &view, false, 0, false, -1, /*keep_counts=*/ci > 0);
        if (rc) break;
        HIPCHK(hipMemcpyAsync(st.pin_results, c->results, (size_t)nr * sizeof(qcat_result), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipEventRecord(st.scanned, c->stream));
        st.res_first = r0; st.res_count = nr;
        t_enqueue += now() - t0;
    }
    const double t_loop = now();
    hipError_t es = hipStreamSynchronize(c->stream);
    hipError_t ec = hipStreamSynchronize(p->copy);
    if (trace)
        fprintf(stderr, "[qcat pipeline] %u reads, %u chunks, %u threads: set-up (pool, pinned and device staging) %.2f ms, staging wait %.2f ms, "
                        "prefix %.2f, compaction %.2f, enqueue %.2f, drain %.2f, total so far %.2f ms\n", n_reads, n_chunks, p->pool->size(),
                t_begin - std::chrono::duration<double, std::milli>(t_entry.time_since_epoch()).count(), t_wait, t_prefix, t_compact, t_enqueue,
                now() - t_loop, now() - t_begin);
    c->last_n_reads = 0;                                 // the context's result buffer holds the last chunk only
    if (rc) return rc;
    if (es != hipSuccess || ec != hipSuccess)
        return set_err(QCAT_ERR_DEVICE, std::string("qcat_scan_batch: ") + hipGetErrorString(es != hipSuccess ? es : ec));
    for (PipeStage& st : p->st) flush_results(st);
    if (counts) {
        std::vector<int64_t> tmp((size_t)hk.n_buckets);
        c->last_buckets = hk.n_buckets;
        if ((rc = qcat_ctx_fetch_counts(c, tmp.data(), hk.n_buckets))) return rc;
        for (size_t i = 0; i < tmp.size(); ++i) counts[i] += tmp[i];
    }
    return 0;
}

// The device work of a host-buffer call -- everything between the upload and the download -- launched kernel by kernel
// (`enqueue`) or replayed as ONE graph.  A call shaped like the one before it (same kit, read count, compacted bases; nothing
// reallocated in between) replays: the launches take no arguments from the host but buffer addresses and sizes, every
// decision that depends on the data (kit vote, job plan, cursors) is made on the device.  The second such call captures
// (hipStreamBeginCapture, thread-local mode; the side streams join the capture through fork_join's events), later ones
// replay (`prep`: host-side state a replay needs as well).  QCAT_HIP_NO_GRAPH=1: always launch kernel by kernel.  Timed
// contexts and the diagnostic switches (they synchronise inside the scan) never capture.  A failure drains the stream before
// it returns (the upload may still be reading the pinned staging).
template <class Enqueue, class Prep>
static int api_graph_run(qcat_ctx* c, qcat_ctx::ApiGraph& G, qcat_kit* kit, KitOnDevice* kd, const qcat_batch* b, uint32_t batch_reads,
                         Enqueue enqueue, Prep prep) {
    const uint32_t n_reads = b->n_reads;
    auto drained = [&](int code) { (void)hipStreamSynchronize(c->stream); return code; };
    static const QcatOpt no_capture[] = {QO_NO_GRAPH, QO_DEBUG_VOTE, QO_BS_TRACE, QO_DEBUG_BINS, QO_DEBUG_REDO};
    bool graph_ok = !c->timing && G.failures < 2 && b->borrowed && !kit->hk.dk.scan_middle && !opt_on(QO_FULL_UPLOAD);      // (whole reads -- --detect-middle -- never replay)
    for (QcatOpt o : no_capture) if (opt_on(o)) graph_ok = false;
    const uint64_t gen_before = g_alloc_gen.load();
    bool done = false;
    int rc = 0;
    if (graph_ok && G.exec && G.kit == kit->serial && G.n_reads == n_reads && G.n_bases == b->n_bases && G.gen == gen_before && G.batch_reads == batch_reads) {
        prep();
        g_jit = kd;
        HIPCHK_DRAIN(c->stream, hipGraphLaunch(G.exec, c->stream));
        ++G.replays;
        done = true;
    } else if (graph_ok && G.prev_kit == kit->serial && G.prev_reads == n_reads && G.prev_bases == b->n_bases && G.prev_gen == gen_before && G.prev_batch == batch_reads) {
        if (G.exec) { (void)hipGraphExecDestroy(G.exec); G.exec = nullptr; }
        hipGraph_t graph = nullptr;
        if (hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal) == hipSuccess) {
            const int erc = enqueue();
            const hipError_t ee = hipStreamEndCapture(c->stream, &graph);
            const bool captured = erc == 0 && ee == hipSuccess && graph != nullptr;
            const bool moved = g_alloc_gen.load() != gen_before;     // (some context allocated meanwhile: not this call's failure)
            if (captured && !moved && hipGraphInstantiate(&G.exec, graph, nullptr, nullptr, 0) == hipSuccess) {
                G.kit = kit->serial; G.n_reads = n_reads; G.n_bases = b->n_bases; G.gen = gen_before; G.batch_reads = batch_reads;
                if (hipGraphLaunch(G.exec, c->stream) == hipSuccess) done = true;
            }
            if (graph) (void)hipGraphDestroy(graph);
            if (!done && !(captured && moved)) {
                ++G.failures;
                (void)hipGetLastError();
                (void)hipStreamSynchronize(c->stream);
                packed_streams_reset(&c->packed);                // (fresh side streams: the old ones may still think they capture)
            }
        } else ++G.failures;
        if (!done) {                                             // (nothing of the capture ran: the plain launches below do the work)
            (void)hipGetLastError();
            if (G.exec) { (void)hipGraphExecDestroy(G.exec); G.exec = nullptr; }
            g_fill_defer = false; g_fill.n = 0;
        }
    }
    if (!done && (rc = enqueue())) return drained(rc);
    G.prev_kit = kit->serial; G.prev_reads = n_reads; G.prev_bases = b->n_bases; G.prev_gen = g_alloc_gen.load(); G.prev_batch = batch_reads;
    return 0;
}

// (ptrs / lens: the reads as one pointer and one length each instead of bases / offsets -- the file loops' one-shot path)
static int scan_debug_impl(qcat_ctx* c, const qcat_kit* ckit,
                           const uint8_t* bases, const uint64_t* offsets, const uint8_t* const* ptrs, const uint64_t* lens, uint32_t n_reads,
                           qcat_result* out, int64_t* counts,
                           qcat_end_trace* traces, int16_t* bc_rows, uint32_t row_stride) {
    if (!c || !ckit || (!offsets && !ptrs) || !out) return set_err(QCAT_ERR_ARG, "null argument");
    qcat_kit* kit = const_cast<qcat_kit*>(ckit);
    if (bc_rows) {
        int need = 0;
        for (int t = 0; t < kit->hk.dk.nt; ++t)
            for (int s = 0; s < 2; ++s) need = std::max(need, kit->hk.dk.tpl[t].sets[s].n);
        if ((int)row_stride < need) return set_err(QCAT_ERR_ARG, "row_stride smaller than the largest barcode set");
    }
    qcat_batch* b = nullptr;
    int rc = batch_upload_windows(c, kit, bases, offsets, n_reads, &b, ptrs, lens);
    if (rc) return rc;
    const bool debug = traces != nullptr || bc_rows != nullptr;
    // (whole reads -- --detect-middle -- never replay a graph: the interior scan sizes buffers from the batch)
    if (!debug && b->borrowed && n_reads && n_reads <= 65536 && !kit->hk.dk.scan_middle && !opt_on(QO_FULL_UPLOAD)) {
        // the reference's library entry -- detect_barcode / detect_barcode_batch with a named kit, down to ONE read per call
        // (qcat/test/test_barcode.py:84, cli.py:504-509) -- is a dozen launches of a few microseconds each: calls of one shape
        // replay them as a graph, like the kit-auto calls (api_graph_run)
        KitOnDevice* kd = nullptr;
        rc = kit_on_device(kit, c->device, &kd);
        if (!rc) rc = api_graph_run(c, c->scan_graph, kit, kd, b, 0, [&] { return scan_resident_impl(c, kit, b, false, 0); }, [] {});
        else (void)hipStreamSynchronize(c->stream);
        if (!rc) { c->last_n_reads = n_reads; c->last_buckets = kit->hk.dk.n_buckets; }      // (a replay does not pass through scan_resident_impl)
    } else rc = scan_resident_impl(c, kit, b, debug, bc_rows ? row_stride : 0);
    if (rc) (void)hipStreamSynchronize(c->stream);          // (the upload may still be reading the context's pinned staging)
    if (!rc) rc = fetch_back(c, out, n_reads, counts, kit->hk.dk.n_buckets);
    if (!rc && debug && n_reads) {
        const int ends = kit->hk.dk.ends == QCAT_ENDS_5P ? 1 : 2;
        const size_t n_ends = (size_t)n_reads * ends;
        std::vector<EndRec> recs(n_ends);
        std::vector<int32_t> tpl(n_ends * 2 * MAX_T);
        hipError_t e = hipMemcpy(recs.data(), c->recs, n_ends * sizeof(EndRec), hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipMemcpy(tpl.data(), c->dbg_tpl, tpl.size() * 4, hipMemcpyDeviceToHost);
        if (e == hipSuccess && bc_rows) e = hipMemcpy(bc_rows, c->dbg_rows, n_ends * 2 * row_stride * 2, hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = set_err(QCAT_ERR_DEVICE, std::string("debug download: ") + hipGetErrorString(e));
        else if (traces)
            for (size_t i = 0; i < n_ends; ++i)
                fill_trace(kit->hk, recs[i], &tpl[(i * 2 + 0) * MAX_T], &tpl[(i * 2 + 1) * MAX_T], &traces[i]);
    }
    qcat_batch_destroy(b);
    return rc;
}

extern "C" int qcat_scan_debug(qcat_ctx* c, const qcat_kit* ckit,
                               const uint8_t* bases, const uint64_t* offsets, uint32_t n_reads,
                               qcat_result* out, int64_t* counts,
                               qcat_end_trace* traces, int16_t* bc_rows, uint32_t row_stride) {
    if (!offsets) return set_err(QCAT_ERR_ARG, "null argument");
    return scan_debug_impl(c, ckit, bases, offsets, nullptr, nullptr, n_reads, out, counts, traces, bc_rows, row_stride);
}

// the vote's device buffer for nb batches: votes [nb][MAX_T], first voters [nb][MAX_T], chosen slots [nb]
static int vote_buffer(qcat_ctx* c, size_t nb) {
    if (c->vote_buf && nb <= c->cap_vote_batches) return 0;
    if (c->vote_buf) { (void)hipFree(c->vote_buf); c->vote_buf = nullptr; c->cap_vote_batches = 0; }
    HIPCHK(q_malloc((void**)&c->vote_buf, nb * (2 * MAX_T * 8 + 4) + 16));
    c->cap_vote_batches = nb;
    return 0;
}

// adapter-only pass over a resident batch + k_vote: per-template votes / first voting read (host arrays of MAX_T)
static int vote_resident(qcat_ctx* c, qcat_kit* kit, const qcat_batch* b, unsigned long long* hv, unsigned long long* hf) {
    for (int t = 0; t < MAX_T; ++t) { hv[t] = 0; hf[t] = ~0ull; }
    int rc = scan_resident_impl(c, kit, b, false, 0, true);
    // (a failed step drains the stream before it returns: the caller's upload may still be reading the pinned staging)
    auto drained = [&](int code) { (void)hipStreamSynchronize(c->stream); return code; };
    if (rc) return drained(rc);
    if (!b->n_reads) return 0;
    KitOnDevice* kd = nullptr;
    if ((rc = kit_on_device(kit, c->device, &kd))) return drained(rc);
    if ((rc = vote_buffer(c, 1))) return drained(rc);
    unsigned long long* d = c->vote_buf;
    HIPCHK(hipMemsetAsync(d, 0, MAX_T * 8, c->stream));
    HIPCHK(hipMemsetAsync(d + MAX_T, 0xFF, MAX_T * 8, c->stream));
    const uint32_t blocks = std::min<uint32_t>((b->n_reads + 255) / 256, 1024);
    hipLaunchKernelGGL(k_vote, dim3(blocks), dim3(256), 0, c->stream, kd->kit, c->recs, b->n_reads, d, d + MAX_T);
    HIPCHK(hipMemcpyAsync(hv, d, MAX_T * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(hf, d + MAX_T, MAX_T * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

extern "C" int qcat_detect_kit(qcat_ctx* c, const qcat_kit* ckit, const uint8_t* bases, const uint64_t* offsets,
                               uint32_t n_reads, int64_t* votes, int64_t* first_read) {
    if (!c || !ckit || !offsets || !votes || !first_read) return set_err(QCAT_ERR_ARG, "null argument");
    qcat_kit* kit = const_cast<qcat_kit*>(ckit);
    if (kit->hk.dk.ends != QCAT_ENDS_BOTH) return set_err(QCAT_ERR_ARG, "qcat_detect_kit needs a kit created with QCAT_ENDS_BOTH");
    const int nt = kit->hk.dk.nt;
    unsigned long long hv[MAX_T], hf[MAX_T];
    qcat_batch* b = nullptr;
    int rc = batch_upload_windows(c, kit, bases, offsets, n_reads, &b);
    if (rc) return rc;
    rc = vote_resident(c, kit, b, hv, hf);
    qcat_batch_destroy(b);
    if (rc) return rc;
    for (int t = 0; t < nt; ++t) {
        votes[t] += (int64_t)hv[t];
        first_read[t] = hf[t] == ~0ull ? (int64_t)n_reads : (int64_t)hf[t];
    }
    return 0;
}

// the kit-auto scan of one batch.  Round 4: ONE host synchronisation -- the vote is counted AND decided on the device
// (k_vote, k_pick_kit), the second pass reads the voted kit slot there (k_adapter_finish), and votes, choice, records and
// counts come back together.  (Round 3 waited for the upload, for the votes and for the records: three round trips in a call
// whose kernels take 0.3 ms.)
static int scan_batch_auto_impl(qcat_ctx* c, const qcat_kit* ckit, const uint8_t* bases, const uint64_t* offsets,
                                const uint8_t* const* ptrs, const uint64_t* lens,
                                uint32_t n_reads, qcat_result* out, int64_t* counts, int32_t* chosen_kit_slot,
                                int64_t* votes, int64_t* first_read, uint32_t batch_reads = 0) {
    // batch_reads > 0 (round 4, the kit-auto file loop): the reads are consecutive batches of that many reads, each votes for
    // a kit of its own (qcat/cli.py:500 calls detect_barcode_batch per batch) -- one adapter pass over all of them, the votes
    // counted and decided per batch (k_vote / k_pick_kit with a batch dimension), the second pass takes every read's kit slot
    // from its batch (k_adapter_finish: slot_span); chosen_kit_slot then has one entry per batch, votes / first_read are unused
    if (!c || !ckit || (n_reads && !offsets && !ptrs) || !out || !chosen_kit_slot) return set_err(QCAT_ERR_ARG, "qcat_scan_batch_auto: null argument");
    qcat_kit* kit = const_cast<qcat_kit*>(ckit);
    const DevKit& hk = kit->hk.dk;
    if (hk.ends != QCAT_ENDS_BOTH) return set_err(QCAT_ERR_ARG, "qcat_scan_batch_auto needs a kit created with QCAT_ENDS_BOTH");
    const uint32_t nb = batch_reads ? (n_reads + batch_reads - 1) / batch_reads : 1u;
    for (uint32_t q = 0; q < std::max(1u, nb); ++q) chosen_kit_slot[q] = -1;
    if (!n_reads) return 0;                                     // (an empty batch: nobody votes, nothing to scan)
    if (batch_reads && (votes || first_read)) return set_err(QCAT_ERR_ARG, "qcat_scan_batches_auto: the per-template votes are reported for one batch only");
    qcat_batch* b = nullptr;
    int rc = batch_upload_windows(c, kit, bases, offsets, n_reads, &b, ptrs, lens);
    if (rc) return rc;
    BatchGuard guard(b);
    // (a failed step drains the stream before it returns: the upload may still be reading the pinned staging)
    auto drained = [&](int code) { (void)hipStreamSynchronize(c->stream); return code; };
    KitOnDevice* kd = nullptr;
    if ((rc = kit_on_device(kit, c->device, &kd))) return drained(rc);
    if ((rc = vote_buffer(c, nb))) return drained(rc);
    unsigned long long* d = c->vote_buf;
    unsigned long long* d_first = d + (size_t)nb * MAX_T;
    int32_t* chosen_dev = reinterpret_cast<int32_t*>(d + 2 * (size_t)nb * MAX_T);
    const uint32_t slot_span = batch_reads ? batch_reads * 2u : 0u;        // read ends per batch (both ends: checked above)
    // the device work of the call, in stream order behind the upload
    auto enqueue = [&]() -> int {
        // pass 1: every template of every kit against both ends (qcat/scanner_base.py:662-678)
        int e = scan_resident_impl(c, kit, b, false, 0, true);
        if (e) return e;
        g_fill_defer = true;
        HIPCHK(packed_fill(d, 0, (size_t)nb * MAX_T * 8, c->stream));
        HIPCHK(packed_fill(d_first, 0xFF, (size_t)nb * MAX_T * 8, c->stream));
        HIPCHK(packed_fill_flush(c->stream));
        const uint32_t per = batch_reads ? batch_reads : n_reads;
        const uint32_t blocks = std::min<uint32_t>((per + 255) / 256, batch_reads ? 64 : 1024);
        hipLaunchKernelGGL(k_vote, dim3(blocks, nb), dim3(256), 0, c->stream, kd->kit, c->recs, n_reads, d, d_first, batch_reads);
        hipLaunchKernelGGL(k_pick_kit, dim3(nb), dim3(64), 0, c->stream, kd->kit, d, d_first, chosen_dev);
        // pass 2: detect_barcode per read with the voted kit's templates (:714-733): the adapter alignments of the vote are
        // still on the device -- only their merge (the kit slot read from chosen_dev), the barcode phase and the finalisation run now
        c->packed.kit_slot_dev = chosen_dev; c->packed.kit_slot_span = slot_span;
        if (opt_on(QO_DEBUG_VOTE)) {                  // (diagnostics: the choice as the device made it, before the second pass)
            unsigned long long dv[2 * MAX_T]; int32_t dc = -7;
            (void)hipStreamSynchronize(c->stream);
            (void)hipMemcpy(dv, d, sizeof dv, hipMemcpyDeviceToHost);
            (void)hipMemcpy(&dc, chosen_dev, 4, hipMemcpyDeviceToHost);
            fprintf(stderr, "[qcat] vote: chosen %d (kit slots %d, templates %d):", dc, hk.n_kit_slots, hk.nt);
            for (int t = 0; t < hk.nt; ++t) fprintf(stderr, " %llu/slot%d", dv[t], hk.tpl[t].kit_slot);
            fprintf(stderr, "\n");
        }
        return scan_resident_impl(c, kit, b, false, 0, false, RESUME_KIT_ON_DEVICE);
    };
    if ((rc = api_graph_run(c, c->api_graph, kit, kd, b, batch_reads, enqueue,
                            [&] { c->packed.kit_slot_dev = chosen_dev; c->packed.kit_slot_span = slot_span; }))) return rc;
    // what comes back, through the context's pinned block: the records, the vote buffer as it lies on the device (votes, first
    // voting reads, chosen slots: one allocation), the counts -- three DMAs
    const size_t bytes_rec = (size_t)n_reads * sizeof(qcat_result), bytes_vote = 2 * (size_t)nb * MAX_T * 8 + (size_t)nb * 4;
    const size_t off_vote = (bytes_rec + 63) / 64 * 64, off_cnt = (off_vote + bytes_vote + 63) / 64 * 64;
    const size_t need_ret = off_cnt + (size_t)hk.n_buckets * 8;
    if (need_ret > c->cap_pin_ret) {
        (void)hipStreamSynchronize(c->stream);
        if (c->pin_ret) (void)hipHostFree(c->pin_ret);
        c->pin_ret = nullptr; c->cap_pin_ret = 0;
        HIPCHK(hipHostMalloc((void**)&c->pin_ret, need_ret + need_ret / 4));
        c->cap_pin_ret = need_ret + need_ret / 4;
    }
    HIPCHK_DRAIN(c->stream, hipMemcpyAsync(c->pin_ret, c->results, bytes_rec, hipMemcpyDeviceToHost, c->stream));
    HIPCHK_DRAIN(c->stream, hipMemcpyAsync(c->pin_ret + off_vote, d, bytes_vote, hipMemcpyDeviceToHost, c->stream));
    if (counts) HIPCHK_DRAIN(c->stream, hipMemcpyAsync(c->pin_ret + off_cnt, c->counts, (size_t)hk.n_buckets * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    memcpy(out, c->pin_ret, bytes_rec);
    const unsigned long long* hv = reinterpret_cast<const unsigned long long*>(c->pin_ret + off_vote);
    const unsigned long long* hf = hv + (size_t)nb * MAX_T;
    const int32_t* chosen = reinterpret_cast<const int32_t*>(hf + (size_t)nb * MAX_T);
    if (!batch_reads)
        for (int t = 0; t < hk.nt; ++t) {
            if (votes) votes[t] += (int64_t)hv[t];
            if (first_read) first_read[t] = hf[t] == ~0ull ? (int64_t)n_reads : (int64_t)hf[t];
        }
    bool all_voted = true;
    for (uint32_t q = 0; q < nb; ++q) { chosen_kit_slot[q] = chosen[q]; all_voted = all_voted && chosen[q] >= 0; }
    if (!all_voted) return set_err(QCAT_ERR_DEVICE, "qcat_scan_batch_auto: no read voted");
    if (counts) {
        const int64_t* tmp = reinterpret_cast<const int64_t*>(c->pin_ret + off_cnt);
        for (size_t i = 0; i < (size_t)hk.n_buckets; ++i) counts[i] += tmp[i];
    }
    return 0;
}

extern "C" int qcat_scan_batch_auto(qcat_ctx* c, const qcat_kit* ckit, const uint8_t* bases, const uint64_t* offsets,
                                    uint32_t n_reads, qcat_result* out, int64_t* counts, int32_t* chosen_kit_slot,
                                    int64_t* votes, int64_t* first_read) {
    return scan_batch_auto_impl(c, ckit, bases, offsets, nullptr, nullptr, n_reads, out, counts, chosen_kit_slot, votes, first_read);
}

extern "C" int qcat_scan_batch_auto_ptrs(qcat_ctx* c, const qcat_kit* ckit, const uint8_t* const* reads, const uint64_t* lengths,
                                         uint32_t n_reads, qcat_result* out, int64_t* counts, int32_t* chosen_kit_slot,
                                         int64_t* votes, int64_t* first_read) {
    if (n_reads && (!reads || !lengths)) return set_err(QCAT_ERR_ARG, "qcat_scan_batch_auto_ptrs: null argument");
    return scan_batch_auto_impl(c, ckit, nullptr, nullptr, reads, lengths, n_reads, out, counts, chosen_kit_slot, votes, first_read);
}

extern "C" int qcat_scan_batches_auto_ptrs(qcat_ctx* c, const qcat_kit* ckit, const uint8_t* const* reads, const uint64_t* lengths,
                                           uint32_t n_reads, uint32_t batch_reads, qcat_result* out, int64_t* counts, int32_t* chosen_kit_slots) {
    if (n_reads && (!reads || !lengths)) return set_err(QCAT_ERR_ARG, "qcat_scan_batches_auto_ptrs: null argument");
    if (!batch_reads) return set_err(QCAT_ERR_ARG, "qcat_scan_batches_auto_ptrs: batch_reads must be positive");
    if (batch_reads >= n_reads)                                 // one batch: the plain call (small batches upload whole reads)
        return scan_batch_auto_impl(c, ckit, nullptr, nullptr, reads, lengths, n_reads, out, counts, chosen_kit_slots, nullptr, nullptr);
    if ((uint64_t)(n_reads + batch_reads - 1) / batch_reads > 65535u) return set_err(QCAT_ERR_ARG, "qcat_scan_batches_auto_ptrs: more than 65535 batches in one call");
    return scan_batch_auto_impl(c, ckit, nullptr, nullptr, reads, lengths, n_reads, out, counts, chosen_kit_slots, nullptr, nullptr, batch_reads);
}

extern "C" int qcat_scan_batch(qcat_ctx* c, const qcat_kit* kit,
                               const uint8_t* bases, const uint64_t* offsets, uint32_t n_reads,
                               qcat_result* out, int64_t* counts) {
    if (!c || !kit || !offsets || !out) return set_err(QCAT_ERR_ARG, "null argument");
    const ReadView rv{bases, offsets, nullptr};
    const int rc = scan_batch_pipelined(c, const_cast<qcat_kit*>(kit), rv, n_reads, out, counts);
    if (rc <= 0) return rc;                           // done (0) or failed (< 0); 1 = take the one-shot path
    return qcat_scan_debug(c, kit, bases, offsets, n_reads, out, counts, nullptr, nullptr, 0);
}

extern "C" int qcat_scan_sequences(qcat_ctx* c, const qcat_kit* ckit, const uint8_t* bases, const uint64_t* offsets,
                                   uint32_t n_seqs, qcat_result* out) {
    if (!c || !ckit || !offsets || !out) return set_err(QCAT_ERR_ARG, "qcat_scan_sequences: null argument");
    qcat_kit* kit = const_cast<qcat_kit*>(ckit);
    for (uint32_t r = 0; r < n_seqs; ++r)
        if (offsets[r + 1] >= offsets[r] && offsets[r + 1] - offsets[r] >= (1ull << 31))
            return set_err(QCAT_ERR_UNSUPPORTED, "qcat_scan_sequences: sequence longer than 2^31 - 1 bases");
    qcat_batch* b = nullptr;
    int rc = qcat_batch_upload(c, bases, offsets, n_seqs, &b);
    if (rc) return rc;
    KitOnDevice* kd = nullptr;
    rc = kit_on_device(kit, c->device, &kd);
    if (!rc) rc = grow(&c->results, &c->cap_reads, (size_t)n_seqs);
    if (!rc && n_seqs) {
        KitPtrs kp{kd->kit, kd->codes, kd->ids, kd->tables};
        const DevKit& hk = kit->hk.dk;
        // one wave per alignment along its anti-diagonals (kernels_tiny.inc; round 5): L + M steps per alignment and every
        // template and barcode of a sequence side by side, where the general kernel walks L x M cells of everything on one
        // lane -- linear gaps and the adapter modes; QCAT_HIP_NO_TINY=1 / simple mode / affine gaps: the general kernel
        int maxb = 1;
        for (int t = 0; t < hk.nt; ++t) for (int s2 = 0; s2 < 2; ++s2) maxb = std::max(maxb, (int)hk.tpl[t].sets[s2].n);
        const bool waves = hk.mode != QCAT_MODE_SIMPLE && hk.gap_open == hk.gap_extend && !opt_on(QO_NO_TINY) && !c->force_generic;
        if (waves) {
            constexpr uint32_t PIECE = 32767;          // (the barcode kernel's grid has (sequence, set) in y: 65535 at most)
            const uint32_t piece = std::min(n_seqs, PIECE);
            rc = grow(&c->recs, &c->cap_recs, (size_t)piece);
            if (!rc) rc = tiny_buffers(c, (size_t)piece, maxb);
            for (uint32_t s0 = 0; !rc && s0 < n_seqs; s0 += PIECE) {
                const uint32_t ns = std::min(PIECE, n_seqs - s0);
                TinyArgs ta{kp, nullptr, nullptr, b->bases, b->offsets + s0, ns, c->recs, c->tiny_tpl, c->tiny_sc, (uint32_t)maxb, nullptr, nullptr, 0, 0};
                hipLaunchKernelGGL(k_tiny_adapter, dim3(ns * (uint32_t)hk.nt), dim3(64), 0, c->stream, ta);
                hipLaunchKernelGGL(k_tiny_decide, dim3((ns + 63) / 64), dim3(64), 0, c->stream, ta);
                hipLaunchKernelGGL(k_tiny_barcode, dim3((uint32_t)maxb, ns * 2), dim3(64), 0, c->stream, ta);
                hipLaunchKernelGGL(k_tiny_select, dim3(ns * 2), dim3(64), 0, c->stream, ta);
                hipLaunchKernelGGL(k_tiny_store_sequences, dim3((ns + 63) / 64), dim3(64), 0, c->stream, ta, c->results + s0);
            }
        } else
        hipLaunchKernelGGL(k_scan_sequences, dim3((n_seqs + GEN_THREADS - 1) / GEN_THREADS), dim3(GEN_THREADS), 0, c->stream,
                           kp, b->bases, b->offsets, n_seqs, c->results);
        c->last_tiny_ends = (waves && !rc) ? n_seqs : 0;
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpyAsync(out, c->results, (size_t)n_seqs * sizeof(qcat_result), hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) rc = set_err(QCAT_ERR_DEVICE, std::string("qcat_scan_sequences: ") + hipGetErrorString(e));
    }
    c->last_n_reads = 0;                               // the context's result buffer no longer holds a batch scan
    qcat_batch_destroy(b);
    return rc;
}

extern "C" int qcat_sg_align(qcat_ctx* c, const uint8_t* queries, const uint64_t* q_offsets, const uint8_t* targets,
                             const uint64_t* t_offsets, uint32_t n, int32_t gap_open, int32_t gap_extend, const int8_t* matrix,
                             int32_t with_stats_and_rule, qcat_alignment* out) {
    const int32_t r1_flag = with_stats_and_rule & QCAT_SG_R1_SCALAR, with_stats = with_stats_and_rule & ~QCAT_SG_R1_SCALAR;
    if (!c || !q_offsets || !t_offsets || !matrix || !out) return set_err(QCAT_ERR_ARG, "qcat_sg_align: null argument");
    if (n == 0) return 0;
    if (gap_open < 0 || gap_extend < 0) return set_err(QCAT_ERR_ARG, "qcat_sg_align: negative gap cost");
    if (with_stats != QCAT_STATS_NONE && with_stats != QCAT_STATS_PARASAIL6 && with_stats != QCAT_STATS_PARASAIL5 && with_stats != QCAT_STATS_ROUND3)
        return set_err(QCAT_ERR_ARG, "qcat_sg_align: with_stats must be one of QCAT_STATS_* (optionally | QCAT_SG_R1_SCALAR)");
    for (uint32_t i = 0; i < n; ++i) {
        if (q_offsets[i + 1] < q_offsets[i] || t_offsets[i + 1] < t_offsets[i]) return set_err(QCAT_ERR_ARG, "offsets must be non-decreasing");
        if (t_offsets[i + 1] - t_offsets[i] > (uint64_t)MAX_TLEN) return set_err(QCAT_ERR_UNSUPPORTED, "qcat_sg_align: target longer than QCAT_MAX_TEMPLATE_LEN");
        if (q_offsets[i + 1] - q_offsets[i] >= (1ull << 31)) return set_err(QCAT_ERR_UNSUPPORTED, "qcat_sg_align: query longer than 2^31 - 1");
    }
    HIPCHK(hipSetDevice(c->device));
    const uint64_t qb = q_offsets[n], tb = t_offsets[n];
    if ((qb && !queries) || (tb && !targets)) return set_err(QCAT_ERR_ARG, "qcat_sg_align: null sequence buffer");
    const uint32_t blocks = (n + GEN_THREADS - 1) / GEN_THREADS;
    const size_t threads = (size_t)blocks * GEN_THREADS;
    DevTemp dq, dqo, dt, dto, dscr, dout;
    HIPCHK(dq.alloc(qb + 1)); HIPCHK(dqo.alloc(((size_t)n + 1) * 8)); HIPCHK(dt.alloc(tb + 1)); HIPCHK(dto.alloc(((size_t)n + 1) * 8));
    HIPCHK(dscr.alloc((with_stats ? 6 : 2) * (size_t)(MAX_TLEN + 1) * threads * 4)); HIPCHK(dout.alloc((size_t)n * sizeof(qcat_alignment)));
    if (qb) HIPCHK(hipMemcpyAsync(dq.p, queries, qb, hipMemcpyHostToDevice, c->stream));
    if (tb) HIPCHK(hipMemcpyAsync(dt.p, targets, tb, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(dqo.p, q_offsets, ((size_t)n + 1) * 8, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(dto.p, t_offsets, ((size_t)n + 1) * 8, hipMemcpyHostToDevice, c->stream));
    SgMatrix m;
    memcpy(m.m, matrix, 49);
    hipLaunchKernelGGL(k_sg_align, dim3(blocks), dim3(GEN_THREADS), 0, c->stream, dq.as<uint8_t>(), dqo.as<uint64_t>(), dt.as<uint8_t>(),
                       dto.as<uint64_t>(), n, (int)gap_open, (int)gap_extend, m, (int)(with_stats | r1_flag), dscr.as<int32_t>(), dout.as<qcat_alignment>());
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, dout.p, (size_t)n * sizeof(qcat_alignment), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

// ------------------------------------------------------------------------------------------
// native FASTQ ingest and egress
// ------------------------------------------------------------------------------------------
#include "fastq_host.inc"
#include "fastq_stream.inc"

// ------------------------------------------------------------------------------------------
// multi-GPU count reduction (RCCL)
// ------------------------------------------------------------------------------------------
#include "comm.inc"
