"""Run-time generation of static-letter kernels for kits that are not in the built-in bundle.

The library ships generated column chains for every template and barcode target of
``resources/kits.json`` (``csrc/static_generated.inc``); a custom kit (``--kit-folder``) would
otherwise run the slower table kernels.  This module emits the same chains for ONE kit descriptor,
compiles them IN PROCESS with hipRTC (``libhiprtc``, against the device-only ``csrc/rtc_prelude.inc``;
``hipcc --genco`` against ``csrc/jit_prelude.inc`` is the fall-back when the hipRTC library is
missing) and hands the code object to the library (``qcat_kit_attach_code``).

    QCAT_AMD_JIT=auto        (default) custom kits start on the table kernels at once, the compile runs
                             on a background thread (seconds to tens of seconds, cached afterwards) and
                             the kit switches to the generated kernels when it is ready
    QCAT_AMD_JIT=1           compile before the kit is first used (benchmarks, steady-state runs)
    QCAT_AMD_JIT=0           never generate code
    QCAT_AMD_JIT_CACHE=dir   where code objects are kept (default ~/.cache/qcat_amd); every cached
                             object is stored with its SHA-256 and verified before it is loaded

When no compiler is available the kit stays on the table kernels and a warning says so (once).
There is no reference counterpart (the reference has one code path); results are identical on
every path, which ``tests/test_jit.py`` checks against the CPU oracle.
"""
import ctypes as C
import hashlib
import logging
import os
import shutil
import subprocess
import tempfile
import threading

from . import abs_plan
from .codes import ASCII_TO_CODE

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
ARCH = "gfx950"
MAX_TEMPLATES = 16
QUAD_MIN_TARGETS = 48            # as tools/gen_static_kernels.py: sets this large also get four-target chains
_LETTER = {0: 0, 1: 1, 2: 2, 3: 3, 4: 4}          # code -> E[] index: A, T, G, C (+ N in templates)


def mode():
    """"off" | "sync" | "auto" from QCAT_AMD_JIT (default auto)."""
    v = os.environ.get("QCAT_AMD_JIT", "auto").strip().lower()
    if v in ("", "0", "off", "no", "false"):
        return "off"
    if v in ("1", "on", "yes", "true", "sync"):
        return "sync"
    return "auto"


def enabled():
    return mode() != "off"


_warned = []


def warn_once(msg):
    if not _warned:
        _warned.append(msg)
        logging.warning(msg)


_rtc = {}


def hiprtc():
    """ctypes handle of libhiprtc (None when the library is not installed)."""
    if "lib" not in _rtc:
        lib = None
        for name in (os.environ.get("QCAT_AMD_HIPRTC"), "libhiprtc.so.7", "libhiprtc.so", "/opt/rocm/lib/libhiprtc.so"):
            if not name:
                continue
            try:
                lib = C.CDLL(name)
                break
            except OSError:
                continue
        if lib is not None:
            vp = C.c_void_p
            lib.hiprtcCreateProgram.argtypes = [C.POINTER(vp), C.c_char_p, C.c_char_p, C.c_int, vp, vp]
            lib.hiprtcCompileProgram.argtypes = [vp, C.c_int, C.POINTER(C.c_char_p)]
            lib.hiprtcGetProgramLogSize.argtypes = [vp, C.POINTER(C.c_size_t)]
            lib.hiprtcGetProgramLog.argtypes = [vp, C.c_char_p]
            lib.hiprtcGetCodeSize.argtypes = [vp, C.POINTER(C.c_size_t)]
            lib.hiprtcGetCode.argtypes = [vp, C.c_char_p]
            lib.hiprtcDestroyProgram.argtypes = [C.POINTER(vp)]
        _rtc["lib"] = lib
    return _rtc["lib"]


def compiler():
    """"hiprtc" | "hipcc" | None: what this process would compile a kit with."""
    if os.environ.get("QCAT_AMD_JIT_COMPILER", "") != "hipcc" and hiprtc() is not None:
        return "hiprtc"
    return "hipcc" if hipcc_path() is not None else None


def hipcc_path():
    cand = os.environ.get("HIPCC") or shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    return cand if os.path.exists(cand) else None


def _codes(seq):
    return [int(ASCII_TO_CODE[ord(c)]) for c in seq]


def _chain(codes):
    """the column chain of tools/gen_static_kernels.py: chunks of four, the next chunk's diagonal term
    formed before the current chunk's writes"""
    e = ["E[%d]" % _LETTER[c] for c in codes]
    out = ["QS_BEGIN(%s)" % e[0]]
    n = len(codes)
    for j in range(0, n, 4):
        k = min(4, n - j)
        inner = e[j + 1:j + k]
        if j + k < n:
            out.append("QS_CHUNK4(%d, %s)" % (j + 1, ", ".join(inner + [e[j + k]])))
        else:
            out.append("QS_LAST%d(%d%s)" % (k, j + 1, "".join(", " + x for x in inner)))
    return " ".join(out)


def _lcp(a, b):
    n = 0
    while n < len(a) and n < len(b) and a[n] == b[n]:
        n += 1
    return n


def _pair_up(targets, flank):
    """greedy pairing of the set's targets by longest common prefix (two targets run in one row pass
    and share their common prefix columns): [(barcode a, barcode b or -1, shared columns)]"""
    idx = range(len(targets))
    cand = sorted(((-_lcp(targets[i], targets[j]), i, j) for i in idx for j in idx if i < j))
    used, pairs = set(), []
    for neg, i, j in cand:
        if i in used or j in used or targets[i] == targets[j]:
            continue
        used.update((i, j))
        pairs.append((i, j, min(-neg, len(targets[i]) - 1)))
    for i in idx:
        if i not in used:
            pairs.append((i, -1, flank))
    return sorted(pairs)


def _abs_ok(descriptor):
    """the rule of kit_prepare.inc (DevKit::abs_ok): the bit-sliced adapter kernels are built on match 5 / mismatch -2 /
    anything against a template N -1 / gap open = extend = 2 and windows of 8..150 bases"""
    d = descriptor.desc
    if not (d.gap_open == 2 and d.gap_extend == 2 and 8 <= d.max_align_length <= 150):
        return False
    m = d.adapter_matrix
    for t in range(4):
        for q in range(4):
            if m[t * 7 + q] != (5 if t == q else -2):
                return False
    return all(m[4 * 7 + q] == -1 for q in range(4))


def _abs_plans(t, sequence):
    """(struct text, entry points, form mask) of template t's bit-sliced adapter plans: two stages when they fit (form 1),
    else four wide stages (form 4); four narrow stages for medium batches when every stage holds <= 13 columns (form 2)"""
    seq = sequence.upper()
    text, entry, forms = [], [], 0
    two = abs_plan.emit_plan("QABJ_T%d" % t, [seq], "template %d" % t)
    if two:
        text.append(two)
        entry.append('extern "C" __global__ void __launch_bounds__(128, 2) qj_abs_%d(qk::AbsArgs a) { qk::abs_body<qabs::QABJ_T%d>(a); }\n' % (t, t))
        forms |= 1
    else:
        wide = abs_plan.emit_multi("QAWJ_T%d" % t, [seq], "template %d (wide stages)" % t, maxc=abs_plan.MW_MAX_COLUMNS)
        if wide:
            text.append(wide)
            entry.append('extern "C" __global__ void __launch_bounds__(256, 2) qj_absw_%d(qk::AbsArgs a) { qk::abs_ms_body<qabs::QAWJ_T%d>(a); }\n' % (t, t))
            forms |= 4
    ms = abs_plan.emit_multi("QAMJ_T%d" % t, [seq], "template %d" % t)
    if ms:
        text.append(ms)
        entry.append('extern "C" __global__ void __launch_bounds__(256, 4) qj_absm_%d(qk::AbsArgs a) { qk::abs_ms_body<qabs::QAMJ_T%d>(a); }\n' % (t, t))
        forms |= 2
    return "".join(text), entry, forms


def generate(descriptor, skip_templates=(), skip_groups=()):
    """(source text, template flags, group flags, pair entries per group) for the templates /
    (template, set) groups of ``descriptor`` that can take static-letter kernels and are not in the
    skip lists"""
    n = int(descriptor.desc.barcode_context_length)
    nsets = 2 if descriptor.mode == "dual" else 1
    tpl_flags = [0] * MAX_TEMPLATES
    grp_flags = [0] * (2 * MAX_TEMPLATES)
    parts = ['#include "jit_prelude.inc"\n', "namespace qk {\n"]
    entry = []
    abs_text = []                                            # bit-sliced adapter plans (namespace qabs, after namespace qk)
    abs_ok = _abs_ok(descriptor) and os.environ.get("QCAT_AMD_JIT_NO_ABS") is None
    entries = [[] for _ in range(2 * MAX_TEMPLATES)]          # per group: (pair case, barcode a, barcode b or -1)
    quads = [[] for _ in range(2 * MAX_TEMPLATES)]            # per group: (quad case, barcodes a, b, c, d, shared column counts)
    for t, lay in enumerate(descriptor.layouts):
        tcodes = _codes(lay.sequence)
        if t not in skip_templates and all(c <= 4 for c in tcodes) and 1 <= len(tcodes) <= 128:
            m = len(tcodes)
            parts.append("struct QACJ_%d { static __device__ __forceinline__ void run(h2 (&h)[%d], h2& carry, h2& left, "
                         "const h2 (&E)[5]) { %s } };\n" % (t, m + 1, _chain(tcodes)))
            entry.append('extern "C" __global__ void __launch_bounds__(qk::PK_WAVES * 64, QS_ADAPTER_WAVES(%d)) '
                         "qj_ad_%d(qk::StaticAdapterArgs a) { qk::adapter_static_body<%d, qk::QACJ_%d>(a); }\n" % (m, t, m, t))
            entry.append('extern "C" __global__ void __launch_bounds__(qk::PK_WAVES * 64, QS_ADAPTER_WAVES(%d)) '
                         "qj_am_%d(qk::MiddleAdapterArgs a) { qk::adapter_middle_body<%d, qk::QACJ_%d>(a); }\n" % (m, t, m, t))
            tpl_flags[t] = 1
            if abs_ok:                                       # bits 1..3: the forms of its bit-sliced plan (qcat_kit_attach_code)
                text, ents, forms = _abs_plans(t, lay.sequence)
                if forms:
                    abs_text.append(text)
                    entry.extend(ents)
                    tpl_flags[t] |= forms << 1
        for s in range(nsets):
            g = t * 2 + s
            bs = lay.get_barcode_set(s)
            if g in skip_groups or not bs:
                continue
            up, dn = lay.get_upstream_context(n, s), lay.get_downstream_context(n, s)
            targets = [_codes(up + b.sequence + dn) for b in bs]
            m = len(targets[0])
            if any(len(x) != m or any(c > 3 for c in x) for x in targets) or not 1 <= m <= 64:
                continue
            u = len(up)
            pairs = _pair_up(targets, u)                     # [(barcode a, barcode b or -1, shared columns)]
            # big sets: two complete pairs per row pass (static_barcode_rows4) -- the flank columns and the per-row
            # work once per four targets; the pairs left over keep their own chains
            quad_list, left = [], []
            if len(targets) >= QUAD_MIN_TARGETS:
                full = [p for p in pairs if p[1] >= 0]
                left = [p for p in pairs if p[1] < 0]
                if len(full) % 2:
                    left.insert(0, full.pop())
                quad_list = [(full[i], full[i + 1]) for i in range(0, len(full), 2)]
            else:
                left = pairs
            for qd, ((a1, a2, ua), (b1, b2, ub)) in enumerate(quad_list):
                t1, t2, t3, t4 = targets[a1], targets[a2], targets[b1], targets[b2]
                u0 = min(_lcp(t1, t3), ua, ub)
                parts.append("struct QSQJ_%d_%d {\n" % (g, qd))
                parts.append("    static __device__ __forceinline__ void pre0(h2 (&h)[%d], h2& carry, h2& left, const h2 (&E)[4]) { %s }\n"
                             % (u0 + 1, _chain(t1[:u0]) if u0 else ""))
                for name, tg, lo, hi in (("prea", t1, u0, ua), ("ta", t1, ua, m), ("tb", t2, ua, m),
                                         ("preb", t3, u0, ub), ("tc", t3, ub, m), ("td", t4, ub, m)):
                    parts.append("    static __device__ __forceinline__ void %s(h2 (&h)[%d], h2& carry, h2& left, const h2 (&E)[4]) { %s }\n"
                                 % (name, hi - lo + 1, _chain(tg[lo:hi]) if hi > lo else ""))
                parts.append("};\n")
                quads[g].append((qd, a1, a2, b1, b2, u0, ua - u0, ub - u0))
            for pr, (ba, bb, up_) in enumerate(left):
                ta, tb = targets[ba], targets[bb if bb >= 0 else ba]
                parts.append("struct QSPJ_%d_%d {\n" % (g, pr))
                parts.append("    static __device__ __forceinline__ void pre(h2 (&h)[%d], h2& carry, h2& left, const h2 (&E)[4]) { %s }\n"
                             % (up_ + 1, _chain(ta[:up_]) if up_ else ""))
                for name, tg in (("ta", ta), ("tb", tb)):
                    parts.append("    static __device__ __forceinline__ void %s(h2 (&h)[%d], h2& carry, h2& left, const h2 (&E)[4]) { %s }\n"
                                 % (name, m - up_ + 1, _chain(tg[up_:])))
                parts.append("};\n")
                entries[g].append((pr, ba, bb))
            parts.append("struct QSGJ_%d {\n    static constexpr int M = %d;\n    static constexpr int HAS_QUADS = %d;\n"
                         "    static __device__ __forceinline__ void run4(int quad, const uint8_t* qbuf, int lane, int Lmax, h2 gL2, "
                         "u32 special, const u32 (&ltr)[4], h2 rowoff, h2 coloff, u32& ra, u32& rb, u32& rc, u32& rd) {\n"
                         "        ra = 0; rb = 0; rc = 0; rd = 0;\n        switch (quad) {\n" % (g, m, 1 if quad_list else 0))
            for qd, a1, a2, b1, b2, u0, da, db in quads[g]:
                parts.append("        case %d: static_barcode_rows4<M, %d, %d, %d, QSQJ_%d_%d>(qbuf, lane, Lmax, gL2, special, ltr, rowoff, coloff, ra, rb, rc, rd); break;\n"
                             % (qd, u0, da, db, g, qd))
            parts.append("        default: break;\n        }\n    }\n"
                         "    static __device__ __forceinline__ void run(int pair, const uint8_t* qbuf, int lane, int Lmax, h2 gL2, "
                         "u32 special, const u32 (&ltr)[4], h2 rowoff, h2 coloff, u32& ra, u32& rb) {\n        ra = 0; rb = 0;\n"
                         "        switch (pair) {\n")
            for pr, (ba, bb, up_) in enumerate(left):
                parts.append("        case %d: static_barcode_rows2<M, %d, QSPJ_%d_%d>(qbuf, lane, Lmax, gL2, special, ltr, rowoff, coloff, ra, rb); break;\n"
                             % (pr, up_, g, pr))
            parts.append("        default: break;\n        }\n    }\n};\n")
            entry.append('extern "C" __global__ void __launch_bounds__(qk::PK_WAVES * 64, %d) '
                         "qj_bc_%d(qk::StaticArgs a) { qk::barcode_static_body<qk::QSGJ_%d>(a); }\n" % (2 if quad_list else 4, g, g))
            grp_flags[g] = 1
            # bit-sliced rows with the letters compiled in (kernels_bitslice.inc): case = barcode index
            shape = _bs_shape(len(up), len(dn), m) if len(targets) <= 128 else None
            if shape:
                rev, pre, own, post = shape
                s1, s0 = _bs_shared_words(targets[0], rev, pre)
                t1, t0 = _bs_trailing_words(targets[0], rev, post)
                parts.append("struct QBSJ_%d {\n    static constexpr int C = %d, KERNEL = QCAT_JIT_BASE + %d, PRE = %d, POST = %d;\n"
                             "    static constexpr unsigned S1 = 0x%Xu, S0 = 0x%Xu, T1 = 0x%Xu, T0 = 0x%Xu;\n"
                             "    static __device__ __forceinline__ void rows(int kase, const BsRowArgs& ra, "
                             "u32 (&h1)[C], u32 (&h0)[C], u32 (&f)[BS_ND]) {\n        switch (kase) {\n" % (g, own, g, pre, post, s1, s0, t1, t0))
                for b, tg in enumerate(targets):
                    w1, w0 = _bs_words(tg, rev, pre, own)
                    parts.append("        case %d: bs_rows_static<C, 0x%XULL, 0x%XULL>(ra, h1, h0, f); break;\n" % (b, w1, w0))
                parts.append("        default: break;\n        }\n    }\n};\n")
                entry.append('extern "C" __global__ void __launch_bounds__(qk::BS_WAVES * 64) '
                             "qj_bs_%d(qk::BsArgs a) { qk::bs_barcode_body<qk::QBSJ_%d>(a); }\n" % (g, g))
                grp_flags[g] |= 2
    parts.append("}  // namespace qk\n")
    if abs_text:
        # the kernel bodies of kernels_abs.inc without the built-in kits' plans, then this kit's plans
        parts.append('#define QCAT_ABS_NO_BUILTIN 1\n#include "kernels_abs.inc"\n'
                     "#define ABS_COPY4(D, S) do { (D)[0] = (S)[0]; (D)[1] = (S)[1]; (D)[2] = (S)[2]; (D)[3] = (S)[3]; } while (0)\n"
                     "namespace qabs {\n" + "".join(abs_text) + "}  // namespace qabs\n")
    return "".join(parts + entry), tpl_flags, grp_flags, entries, [[q[:5] for q in g] for g in quads]


BS_C_MIN, BS_C_MAX = 20, 48          # kit.h


BS_POSTS = (11, 8, 7, 6, 4)          # kit.h: bs_post_of


def _bs_shape(uplen, downlen, m):
    """(reversed, shared columns, own columns, trailing columns) of a set on the bit-sliced kernels, or None -- the rule of
    kit_prepare.inc: the longer context leads, 11 / 8 / 4 / 0 of its columns are shared; 11 / 8 / 7 / 6 / 4 / 0 columns of the
    other context go through the reversed DP (csrc/bs_core.h)"""
    rev = downlen > uplen
    lead, trail = (downlen, uplen) if rev else (uplen, downlen)
    pre = 11 if lead >= 11 else (8 if lead >= 8 else (4 if lead >= 4 else 0))
    post = next((q for q in BS_POSTS if q <= trail and m - pre - q >= BS_C_MIN), 0)
    own = m - pre - post
    if not (BS_C_MIN <= own <= BS_C_MAX and m <= 64):
        return None
    return rev, pre, own, post


def _bs_words(codes, rev, pre, own):
    """letter bit words of the own columns in the order the kernel walks them (bit j = own column j)"""
    t = codes[::-1] if rev else codes
    w1 = w0 = 0
    for j, c in enumerate(t[pre:pre + own]):
        w1 |= ((c >> 1) & 1) << j
        w0 |= (c & 1) << j
    return w1, w0


def _bs_trailing_words(codes, rev, post):
    """letter bit words of the trailing columns in the order the reversed DP walks them (bit j = the last column but j)"""
    t = codes[::-1] if rev else codes
    w1 = w0 = 0
    for j in range(post):
        c = t[len(t) - 1 - j]
        w1 |= ((c >> 1) & 1) << j
        w0 |= (c & 1) << j
    return w1, w0


def _bs_shared_words(codes, rev, pre):
    """letter bit words of the shared (leading context) columns: bit j = shared column j"""
    t = codes[::-1] if rev else codes
    w1 = w0 = 0
    for j, c in enumerate(t[:pre]):
        w1 |= ((c >> 1) & 1) << j
        w0 |= (c & 1) << j
    return w1, w0


def _prelude_digest():
    h = hashlib.sha1()
    for name in sorted(os.listdir(CSRC)):
        if name.endswith((".inc", ".h")) and name != "static_generated.inc":
            with open(os.path.join(CSRC, name), "rb") as fh:
                h.update(name.encode())
                h.update(fh.read())
    # kit.h includes the ABI header (struct layouts, enums, QCAT_MAX_*): part of every code object's layout
    abi = os.path.join(os.path.dirname(os.path.dirname(CSRC)), "include", "qcat_hip.h")
    try:
        with open(abi, "rb") as fh:
            h.update(b"qcat_hip.h")
            h.update(fh.read())
    except OSError:
        pass
    return h.hexdigest()


_toolchain = {}


def _toolchain_version(how):
    """version string of the compiler a code object comes from (part of the cache key)"""
    if how not in _toolchain:
        ver = ""
        try:
            if how == "hiprtc":
                major, minor = C.c_int(0), C.c_int(0)
                if hiprtc().hiprtcVersion(C.byref(major), C.byref(minor)) == 0:
                    ver = "hiprtc-%d.%d" % (major.value, minor.value)
            else:
                out = subprocess.run([hipcc_path(), "--version"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
                ver = hashlib.sha1(out.stdout).hexdigest()
        except (OSError, AttributeError, TypeError):
            ver = ""
        _toolchain[how] = ver
    return _toolchain[how]


def _compile_hiprtc(source):
    rtc = hiprtc()
    prog = C.c_void_p()
    src = source.replace('#include "jit_prelude.inc"', '#include "rtc_prelude.inc"').encode()
    if rtc.hiprtcCreateProgram(C.byref(prog), src, b"qcat_kit.hip", 0, None, None) != 0:
        raise RuntimeError("qcat_amd.jit: hiprtcCreateProgram failed")
    try:
        opts = [b"--offload-arch=" + ARCH.encode(), b"-O3", b"-std=c++17", b"-w", b"-I" + CSRC.encode()]
        arr = (C.c_char_p * len(opts))(*opts)
        rc = rtc.hiprtcCompileProgram(prog, len(opts), arr)
        if rc != 0:
            n = C.c_size_t()
            rtc.hiprtcGetProgramLogSize(prog, C.byref(n))
            log = C.create_string_buffer(max(1, n.value))
            rtc.hiprtcGetProgramLog(prog, log)
            raise RuntimeError("qcat_amd.jit: hipRTC failed (%d):\n%s" % (rc, log.value.decode(errors="replace")[-4000:]))
        n = C.c_size_t()
        rtc.hiprtcGetCodeSize(prog, C.byref(n))
        blob = C.create_string_buffer(n.value)
        if rtc.hiprtcGetCode(prog, blob) != 0 or n.value == 0:
            raise RuntimeError("qcat_amd.jit: hiprtcGetCode failed")
        return blob.raw
    finally:
        rtc.hiprtcDestroyProgram(C.byref(prog))


def _compile_hipcc(source):
    hipcc = hipcc_path()
    tmp = tempfile.mkdtemp(prefix="qcat_jit_")
    try:
        src = os.path.join(tmp, "kit.hip")
        out = os.path.join(tmp, "kit.hsaco")
        with open(src, "w") as fh:
            fh.write(source)
        cmd = [hipcc, "--genco", "--offload-arch=" + ARCH, "-O3", "-std=c++17", "-w", "-I", CSRC, src, "-o", out]
        proc = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        if proc.returncode != 0 or not os.path.exists(out):
            raise RuntimeError("qcat_amd.jit: hipcc failed:\n" + proc.stdout.decode(errors="replace")[-4000:])
        with open(out, "rb") as fh:
            return fh.read()
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def cache_dir():
    return os.environ.get("QCAT_AMD_JIT_CACHE") or os.path.join(os.path.expanduser("~"), ".cache", "qcat_amd")


def _cache_load(path):
    """the cached code object, or None when it is absent or does not match its recorded SHA-256.  The digest sits
    beside the blob in the same user-writable directory, so this catches truncation and corruption (a half-written
    file must never reach hipModuleLoadData), not tampering by someone who can write there."""
    try:
        with open(path, "rb") as fh:
            blob = fh.read()
        with open(path + ".sha256") as fh:
            want = fh.read().strip()
    except OSError:
        return None
    return blob if hashlib.sha256(blob).hexdigest() == want else None


def _cache_store(path, blob):
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        tmp = path + ".tmp%d" % os.getpid()
        with open(tmp, "wb") as fh:
            fh.write(blob)
        os.replace(tmp, path)
        with open(tmp, "w") as fh:
            fh.write(hashlib.sha256(blob).hexdigest() + "\n")
        os.replace(tmp, path + ".sha256")
    except OSError:
        pass                                        # read-only home: compile again next time


def compile_source(source):
    """code object bytes for ``source`` (cached by content of the source and of the kernel headers)"""
    how = compiler()
    if how is None:
        raise RuntimeError("qcat_amd.jit: neither libhiprtc nor hipcc found (QCAT_AMD_HIPRTC / HIPCC); "
                           "cannot generate kernels for this kit")
    # the key covers everything the code object's layout depends on: the generated source, the kernel headers AND the
    # ABI header they include (struct / enum / MAX_* constants), the compile route (hipcc and hipRTC use different
    # preludes) and the toolchain version
    key = hashlib.sha1((source + _prelude_digest() + ARCH + how + _toolchain_version(how)).encode()).hexdigest()
    path = os.path.join(cache_dir(), key + ".hsaco")
    blob = _cache_load(path)
    if blob is not None:
        return blob
    blob = _compile_hiprtc(source) if how == "hiprtc" else _compile_hipcc(source)
    _cache_store(path, blob)
    return blob


def needs_code(info):
    """does a kit with this ``describe()`` leave templates / groups on the table kernels?"""
    return bool(info["packed"]) and not (info["n_static_templates"] == info["n_templates"]
                                         and info["n_static_groups"] == info["n_groups"])


def attach(native_kit, handle=None):
    """generate, compile and attach kernels for whatever the built-in registry left on the table
    kernels; returns the kit's new ``describe()`` (unchanged when there was nothing to do).
    ``handle``: attach to this (not yet used) ``qcat_kit*`` instead of the kit's current one."""
    info = native_kit.describe()
    if not needs_code(info):
        return info
    source, tpl_flags, grp_flags, entries, quads = generate(native_kit.descriptor)
    if not any(tpl_flags) and not any(grp_flags):
        return info
    blob = compile_source(source)
    hip = native_kit.hip
    tf = (C.c_int32 * MAX_TEMPLATES)(*tpl_flags)
    gf = (C.c_int32 * (2 * MAX_TEMPLATES))(*grp_flags)
    offs, flat = [0], []
    for g in range(2 * MAX_TEMPLATES):
        for pr, ba, bb in entries[g]:
            flat.extend((pr, ba, bb))
        offs.append(len(flat) // 3)
    po = (C.c_int32 * len(offs))(*offs)
    pe = (C.c_int32 * max(1, len(flat)))(*flat)
    qoffs, qflat = [0], []
    for g in range(2 * MAX_TEMPLATES):
        for q in quads[g]:
            qflat.extend(q)
        qoffs.append(len(qflat) // 5)
    qo = (C.c_int32 * len(qoffs))(*qoffs)
    qe = (C.c_int32 * max(1, len(qflat)))(*qflat)
    hip.check(hip.lib.qcat_kit_attach_code_quads(handle if handle is not None else native_kit.handle,
                                                 blob, len(blob), tf, gf, po, pe, qo, qe))
    return native_kit.describe(handle)


def attach_in_background(native_kit):
    """QCAT_AMD_JIT=auto: the kit is usable at once on the table kernels; a daemon thread compiles
    its kernels, prepares a second ``qcat_kit*`` with the code attached and publishes it through
    ``native_kit.upgrade()`` -- scans issued after that run the generated kernels."""
    if not needs_code(native_kit.describe()):
        return None

    def work():
        try:
            handle = native_kit.new_handle()
            try:
                attach(native_kit, handle)
            except Exception:
                native_kit.hip.lib.qcat_kit_destroy(handle)
                raise
            native_kit.upgrade(handle)
        except Exception as exc:                     # noqa: BLE001 -- never take the scan down with it
            warn_once("qcat_amd: run-time kernel generation failed, the kit stays on the table kernels: %s" % exc)

    th = threading.Thread(target=work, name="qcat-jit", daemon=True)
    th.start()
    return th
