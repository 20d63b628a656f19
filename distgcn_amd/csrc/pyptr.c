/* _pyptr: the one piece of CPython glue next to the ctypes binding.  libdgcn.so's ingestion entry
 * (dgcn_pack_batch, include/dgcn.h) takes tables of host pointers, one per graph; collecting 1 500 addresses
 * through ndarray.ctypes / __array_interface__ costs ~1 ms per 500-graph batch in the interpreter - more than the
 * packing and the kernel together.  addresses() walks a sequence of buffer objects (NumPy arrays) through the
 * buffer protocol and fills the table in ~50 ns per item.  No computation happens here. */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <stdint.h>
#include <string.h>

/* addresses(seq, addr_out, count_out_or_None, kind) -> itemsize
 *   seq        sequence of C-contiguous buffers (None allowed: address 0, count 0)
 *   addr_out   writable buffer of len(seq) uint64
 *   count_out  None or writable buffer of len(seq) int64: element counts
 *   kind       0: signed integers, all of one itemsize (4 or 8), which is returned
 *              4 / 8: signed integers of exactly that itemsize;  64: float64 */
static PyObject* addresses(PyObject* self, PyObject* args) {
    PyObject *seq, *addr_obj, *cnt_obj;
    int kind;
    if (!PyArg_ParseTuple(args, "OOOi", &seq, &addr_obj, &cnt_obj, &kind)) return NULL;
    PyObject* fast = PySequence_Fast(seq, "addresses(): first argument must be a sequence");
    if (!fast) return NULL;
    const Py_ssize_t n = PySequence_Fast_GET_SIZE(fast);
    Py_buffer addr, cnt;
    memset(&cnt, 0, sizeof(cnt));
    if (PyObject_GetBuffer(addr_obj, &addr, PyBUF_WRITABLE | PyBUF_SIMPLE) < 0) { Py_DECREF(fast); return NULL; }
    int have_cnt = cnt_obj != Py_None;
    if (have_cnt && PyObject_GetBuffer(cnt_obj, &cnt, PyBUF_WRITABLE | PyBUF_SIMPLE) < 0) {
        PyBuffer_Release(&addr); Py_DECREF(fast); return NULL;
    }
    PyObject* result = NULL;
    if (addr.len < n * 8 || (have_cnt && cnt.len < n * 8)) {
        PyErr_SetString(PyExc_ValueError, "addresses(): output buffers are too small");
        goto done;
    }
    {
        uint64_t* a = (uint64_t*)addr.buf;
        int64_t* c = have_cnt ? (int64_t*)cnt.buf : NULL;
        int itemsize = (kind == 4 || kind == 8) ? kind : (kind == 64 ? 8 : 0);
        for (Py_ssize_t i = 0; i < n; ++i) {
            PyObject* item = PySequence_Fast_GET_ITEM(fast, i);
            if (item == Py_None) { a[i] = 0; if (c) c[i] = 0; continue; }
            Py_buffer v;
            if (PyObject_GetBuffer(item, &v, PyBUF_FORMAT | PyBUF_C_CONTIGUOUS) < 0) goto done;
            const char* f = v.format ? v.format : "B";
            while (*f == '@' || *f == '=' || *f == '<') ++f;
            const int is_float = (*f == 'd');
            const int is_int = (*f == 'i' || *f == 'l' || *f == 'q');
            int ok = (kind == 64) ? (is_float && v.itemsize == 8) : (is_int && (v.itemsize == 4 || v.itemsize == 8));
            if (ok && kind != 64) {
                if (itemsize == 0) itemsize = (int)v.itemsize;
                ok = v.itemsize == itemsize;
            }
            if (!ok) {
                PyErr_Format(PyExc_TypeError, "addresses(): item %zd has format '%s' / itemsize %zd, expected %s",
                             i, v.format ? v.format : "?", v.itemsize,
                             kind == 64 ? "float64" : "signed integers of one width (int32 or int64)");
                PyBuffer_Release(&v);
                goto done;
            }
            a[i] = (uint64_t)(uintptr_t)v.buf;
            if (c) c[i] = (int64_t)(v.len / v.itemsize);
            PyBuffer_Release(&v);
        }
        result = PyLong_FromLong(itemsize ? itemsize : 4);
    }
done:
    if (have_cnt) PyBuffer_Release(&cnt);
    PyBuffer_Release(&addr);
    Py_DECREF(fast);
    return result;
}

static PyMethodDef methods[] = {{"addresses", addresses, METH_VARARGS, "fill a table of buffer addresses"}, {NULL, NULL, 0, NULL}};
static struct PyModuleDef moddef = {PyModuleDef_HEAD_INIT, "_pyptr", "pointer-table helper of distgcn_amd", -1, methods};
PyMODINIT_FUNC PyInit__pyptr(void) { return PyModule_Create(&moddef); }
