/* pyglue.c -> qcat_amd/_pyglue.so: two conversions at the Python boundary of the drop-in, in C (CPython API).
 *
 * The reference's batch entry point takes a LIST OF STR and returns a LIST OF DICTS (qcat/scanner_base.py:714-733,
 * build_return_dict :362-390).  Around a 4000-read call whose kernels take 0.3 ms, making the list one byte buffer
 * (join + encode: 0.8 ms) and making 4000 seven-key dicts from the record array in a Python loop (1.0 ms) were two thirds
 * of the call.  Here:
 *   read_views(reads)                 -> (pointers, lengths): one pointer and one length per read, taken from the str /
 *                                        bytes objects themselves (no copy; the native library reads the windows it needs
 *                                        straight from them: qcat_scan_batch_auto_ptrs).  None when an element is not an
 *                                        ASCII str, bytes or None -- the caller then packs the list as before.
 *   records_to_dicts(recs, bars, ads) -> the list of result dicts of a record array (epi2me records; a record with a second
 *                                        barcode -- dual mode names a barcode per pair -- makes it return None: Python loop).
 * Host glue only: no alignment arithmetic here, and the package works without it (pure-Python fall-backs, native.py). */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <stdint.h>
#include <string.h>

static PyObject* read_views(PyObject* self, PyObject* arg) {
    (void)self;
    if (!PyList_Check(arg)) Py_RETURN_NONE;
    const Py_ssize_t n = PyList_GET_SIZE(arg);
    PyObject* ptrs = PyBytes_FromStringAndSize(NULL, n * 8);
    PyObject* lens = PyBytes_FromStringAndSize(NULL, n * 8);
    if (!ptrs || !lens) { Py_XDECREF(ptrs); Py_XDECREF(lens); return NULL; }
    uint64_t* p = (uint64_t*)PyBytes_AS_STRING(ptrs);
    uint64_t* l = (uint64_t*)PyBytes_AS_STRING(lens);
    for (Py_ssize_t i = 0; i < n; ++i) {
        PyObject* o = PyList_GET_ITEM(arg, i);
        if (o == Py_None) { p[i] = 0; l[i] = 0; }
        else if (PyUnicode_Check(o)) {
            if (PyUnicode_READY(o) < 0 || !PyUnicode_IS_COMPACT_ASCII(o)) { Py_DECREF(ptrs); Py_DECREF(lens); PyErr_Clear(); Py_RETURN_NONE; }
            p[i] = (uint64_t)(uintptr_t)PyUnicode_1BYTE_DATA(o); l[i] = (uint64_t)PyUnicode_GET_LENGTH(o);
        } else if (PyBytes_Check(o)) {
            p[i] = (uint64_t)(uintptr_t)PyBytes_AS_STRING(o); l[i] = (uint64_t)PyBytes_GET_SIZE(o);
        } else { Py_DECREF(ptrs); Py_DECREF(lens); Py_RETURN_NONE; }
    }
    PyObject* out = PyTuple_Pack(2, ptrs, lens);
    Py_DECREF(ptrs); Py_DECREF(lens);
    return out;
}

/* qcat_result, include/qcat_hip.h (24 bytes, little-endian) */
typedef struct { int16_t barcode_idx, barcode2_idx, adapter_idx, exit_status; int32_t adapter_end, trim5p, trim3p; int16_t raw_score, score_den; } rec_t;

static PyObject *k_barcode, *k_score, *k_adapter, *k_end, *k_t5, *k_t3, *k_exit;
static PyObject* dict_template;          /* the seven keys in the reference's order, values None: a result dict starts as a copy (public API,
                                          one allocation of the right size; PyDict_New() grows at the sixth key) */

enum { SCORE_SLOTS = 1024, LONG_SLOTS = 16384 };
typedef struct { uint32_t key; PyObject* obj; } score_slot;
static score_slot score_cache[SCORE_SLOTS];
static PyObject* long_cache[LONG_SLOTS];      /* 0 .. LONG_SLOTS - 1, made on first use */
static PyObject* cached_long(long v) {
    if (v < 0 || v >= LONG_SLOTS) return PyLong_FromLong(v);
    PyObject* o = long_cache[v];
    if (!o) { o = PyLong_FromLong(v); if (!o) return NULL; long_cache[v] = o; }
    Py_INCREF(o);
    return o;
}

static PyObject* records_to_dicts(PyObject* self, PyObject* args) {
    (void)self;
    Py_buffer view;
    PyObject *bars, *ads;
    if (!PyArg_ParseTuple(args, "y*OO", &view, &bars, &ads)) return NULL;
    PyObject* out = NULL;
    if (!PyList_Check(bars) || !PyList_Check(ads) || view.len % (Py_ssize_t)sizeof(rec_t) != 0) {
        PyBuffer_Release(&view);
        PyErr_SetString(PyExc_TypeError, "records_to_dicts(record bytes, barcode table (list of lists), adapter table (list))");
        return NULL;
    }
    const Py_ssize_t n = view.len / (Py_ssize_t)sizeof(rec_t);
    const rec_t* r = (const rec_t*)view.buf;
    for (Py_ssize_t i = 0; i < n; ++i)
        if (r[i].barcode2_idx >= 0) { PyBuffer_Release(&view); Py_RETURN_NONE; }      /* dual records: a Barcode per pair, the Python loop */
    out = PyList_New(n);
    if (!out) { PyBuffer_Release(&view); return NULL; }
    const Py_ssize_t n_ads = PyList_GET_SIZE(ads), n_rows = PyList_GET_SIZE(bars);
    for (Py_ssize_t i = 0; i < n; ++i) {
        const Py_ssize_t a = (Py_ssize_t)r[i].adapter_idx + 1, b = (Py_ssize_t)r[i].barcode_idx + 1;
        PyObject* barcode = Py_None;
        if (b > 0) {                                     /* the table's row 0 / column 0 are None (index -1) */
            PyObject* row = (a >= 0 && a < n_rows) ? PyList_GET_ITEM(bars, a) : NULL;
            if (!row || !PyList_Check(row) || b >= PyList_GET_SIZE(row)) { PyErr_SetString(PyExc_IndexError, "record index outside the kit tables"); goto fail; }
            barcode = PyList_GET_ITEM(row, b);
        }
        if (a < 0 || a >= n_ads) { PyErr_SetString(PyExc_IndexError, "record index outside the kit tables"); goto fail; }
        PyObject* adapter = PyList_GET_ITEM(ads, a);
        /* the same IEEE double expression as qcat/scanner_base.py:119: raw * 100.0 / (1.0 * den) */
        const double den = r[i].score_den > 1 ? (double)r[i].score_den : 1.0;
        const double score = b > 0 ? (double)r[i].raw_score * 100.0 / (1.0 * den) : 0.0;
        PyObject* d = PyDict_Copy(dict_template);
        /* a score is raw * 100 / den of two small integers: a few dozen distinct floats per kit, kept (a float is immutable);
           read-length sized integers (trim3p, exit codes) likewise -- CPython itself keeps only -5..256 */
        PyObject* v_score;
        if (b <= 0) { v_score = PyFloat_FromDouble(0.0); }      /* (no barcode: 0.0 whatever the record's raw fields hold) */
        else {
            const uint32_t key = ((uint32_t)(uint16_t)r[i].raw_score << 16) | (uint16_t)r[i].score_den;
            score_slot* sl = &score_cache[(key ^ (key >> 11)) & (SCORE_SLOTS - 1)];
            if (sl->obj && sl->key == key) { v_score = sl->obj; Py_INCREF(v_score); }
            else {
                v_score = PyFloat_FromDouble(score);
                if (v_score) { Py_XDECREF(sl->obj); sl->obj = v_score; sl->key = key; Py_INCREF(v_score); }
            }
        }
        PyObject* v_end = cached_long(r[i].adapter_end);
        PyObject* v_t5 = cached_long(r[i].trim5p);
        PyObject* v_t3 = cached_long(r[i].trim3p);
        PyObject* v_exit = cached_long(r[i].exit_status);
        int bad = !d || !v_score || !v_end || !v_t5 || !v_t3 || !v_exit;
        if (!bad)
            bad = PyDict_SetItem(d, k_barcode, barcode) < 0 || PyDict_SetItem(d, k_score, v_score) < 0 || PyDict_SetItem(d, k_adapter, adapter) < 0 ||
                  PyDict_SetItem(d, k_end, v_end) < 0 || PyDict_SetItem(d, k_t5, v_t5) < 0 || PyDict_SetItem(d, k_t3, v_t3) < 0 ||
                  PyDict_SetItem(d, k_exit, v_exit) < 0;
        Py_XDECREF(v_score); Py_XDECREF(v_end); Py_XDECREF(v_t5); Py_XDECREF(v_t3); Py_XDECREF(v_exit);
        if (bad) { Py_XDECREF(d); goto fail; }
        PyList_SET_ITEM(out, i, d);
    }
    PyBuffer_Release(&view);
    return out;
fail:
    PyBuffer_Release(&view);
    Py_DECREF(out);
    return NULL;
}

static PyMethodDef methods[] = {
    {"read_views", read_views, METH_O, "list of str / bytes / None -> (pointer bytes, length bytes), or None"},
    {"records_to_dicts", records_to_dicts, METH_VARARGS, "record bytes, barcode table, adapter table -> list of result dicts, or None"},
    {NULL, NULL, 0, NULL}
};
static struct PyModuleDef module = {PyModuleDef_HEAD_INIT, "_pyglue", "list <-> buffer conversions of the qcat_amd drop-in", -1, methods, NULL, NULL, NULL, NULL};

PyMODINIT_FUNC PyInit__pyglue(void) {
    k_barcode = PyUnicode_InternFromString("barcode"); k_score = PyUnicode_InternFromString("barcode_score");
    k_adapter = PyUnicode_InternFromString("adapter"); k_end = PyUnicode_InternFromString("adapter_end");
    k_t5 = PyUnicode_InternFromString("trim5p"); k_t3 = PyUnicode_InternFromString("trim3p");
    k_exit = PyUnicode_InternFromString("exit_status");
    if (!k_barcode || !k_score || !k_adapter || !k_end || !k_t5 || !k_t3 || !k_exit) return NULL;
    dict_template = PyDict_New();
    if (!dict_template) return NULL;
    {
        PyObject* keys[7] = {k_barcode, k_score, k_adapter, k_end, k_t5, k_t3, k_exit};
        for (int i = 0; i < 7; ++i)
            if (PyDict_SetItem(dict_template, keys[i], Py_None) < 0) return NULL;
    }
    return PyModule_Create(&module);
}
