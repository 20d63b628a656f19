/* _pyptr: the CPython glue next to the ctypes binding.  libdgcn.so's ingestion entry
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

/* solve_lists(submit_addr, result_addr, handle_addr, indptrs, indices, weights_or_None)
 *   -> (state, totals, rounds, scores_or_None, status_bits, num_nodes_per_graph)   [bytearrays]   or   int error code
 * One interpreter call around dgcn_host_solver_submit + dgcn_host_solver_result (include/dgcn.h) for the per-graph API
 * calls: the reference's agents are called with one graph at a time, and a dozen NumPy / ctypes calls around a 100 us
 * kernel cost half as much again.  The two entry points arrive as addresses (this module does not link the library).
 * Same checks as the interpreter path: contiguous buffers, indptr / indices signed integers of ONE width (TypeError),
 * float64 weights (TypeError), lengths that match (ValueError).  No computation happens here. */
typedef int (*submit_fn)(void*, const void* const*, const void* const*, const double* const*, const int32_t*, int32_t, int32_t);
typedef int (*result_fn)(void*, int32_t, const uint8_t**, const double**, const int32_t**, const float**, int32_t*, int32_t*, int32_t*);

#define SOLVE_MAX 64

static int int_format(const Py_buffer* v) {
    const char* f = v->format ? v->format : "B";
    while (*f == '@' || *f == '=' || *f == '<') ++f;
    return (*f == 'i' || *f == 'l' || *f == 'q') && (v->itemsize == 4 || v->itemsize == 8);
}

static PyObject* solve_lists(PyObject* self, PyObject* args) {
    unsigned long long submit_addr, result_addr, handle_addr;
    PyObject *ip_seq, *ix_seq, *w_seq;
    if (!PyArg_ParseTuple(args, "KKKOOO", &submit_addr, &result_addr, &handle_addr, &ip_seq, &ix_seq, &w_seq)) return NULL;
    PyObject* ip_fast = PySequence_Fast(ip_seq, "solve_lists(): indptrs must be a sequence");
    if (!ip_fast) return NULL;
    PyObject* ix_fast = PySequence_Fast(ix_seq, "solve_lists(): indices must be a sequence");
    if (!ix_fast) { Py_DECREF(ip_fast); return NULL; }
    PyObject* w_fast = NULL;
    if (w_seq != Py_None) {
        w_fast = PySequence_Fast(w_seq, "solve_lists(): weights must be a sequence or None");
        if (!w_fast) { Py_DECREF(ip_fast); Py_DECREF(ix_fast); return NULL; }
    }
    const Py_ssize_t n = PySequence_Fast_GET_SIZE(ip_fast);
    Py_buffer bufs[3 * SOLVE_MAX];
    int held = 0;
    const void* ap[SOLVE_MAX];
    const void* ai[SOLVE_MAX];
    const double* aw[SOLVE_MAX];
    int32_t nn[SOLVE_MAX];
    PyObject* out = NULL;
    int isz = 0;
    if (n > SOLVE_MAX || PySequence_Fast_GET_SIZE(ix_fast) != n || (w_fast && PySequence_Fast_GET_SIZE(w_fast) != n)) {
        PyErr_SetString(PyExc_ValueError, "indptr / indices / weights lists differ in length (or hold more than 64 graphs)");
        goto done;
    }
    for (Py_ssize_t i = 0; i < n; ++i) {
        Py_buffer* p = &bufs[held];
        if (PyObject_GetBuffer(PySequence_Fast_GET_ITEM(ip_fast, i), p, PyBUF_FORMAT | PyBUF_C_CONTIGUOUS) < 0) goto done;
        ++held;
        Py_buffer* x = &bufs[held];
        if (PyObject_GetBuffer(PySequence_Fast_GET_ITEM(ix_fast, i), x, PyBUF_FORMAT | PyBUF_C_CONTIGUOUS) < 0) goto done;
        ++held;
        if (isz == 0 && int_format(p)) isz = (int)p->itemsize;
        if (!int_format(p) || !int_format(x) || p->itemsize != isz || x->itemsize != isz) {
            PyErr_Format(PyExc_TypeError, "graph %zd: indptr / indices must be int32 or int64, one width for the whole call", i);
            goto done;
        }
        const Py_ssize_t rows = p->len / p->itemsize - 1;
        if (rows < 0) { PyErr_SetString(PyExc_ValueError, "an indptr array is empty"); goto done; }
        const long long last = isz == 4 ? (long long)((const int32_t*)p->buf)[rows] : (long long)((const int64_t*)p->buf)[rows];
        if (last != (long long)(x->len / x->itemsize)) {
            PyErr_SetString(PyExc_ValueError, "an indices array does not match its indptr");
            goto done;
        }
        if (rows > 0x7fffffff) { PyErr_SetString(PyExc_ValueError, "too many vertices"); goto done; }
        ap[i] = p->buf;
        ai[i] = x->buf;
        nn[i] = (int32_t)rows;
        aw[i] = NULL;
        if (w_fast) {
            Py_buffer* w = &bufs[held];
            if (PyObject_GetBuffer(PySequence_Fast_GET_ITEM(w_fast, i), w, PyBUF_FORMAT | PyBUF_C_CONTIGUOUS) < 0) goto done;
            ++held;
            const char* f = w->format ? w->format : "B";
            while (*f == '@' || *f == '=' || *f == '<') ++f;
            if (*f != 'd' || w->itemsize != 8) {
                PyErr_Format(PyExc_TypeError, "graph %zd: weights must be float64", i);
                goto done;
            }
            if (w->len / 8 != rows) {
                PyErr_SetString(PyExc_ValueError, "a weights array does not match its graph's vertex count");
                goto done;
            }
            aw[i] = (const double*)w->buf;
        }
    }
    {
        void* h = (void*)(uintptr_t)handle_addr;
        int slot, rc = 0;
        const uint8_t* state = NULL;
        const double* totals = NULL;
        const int32_t* rounds = NULL;
        const float* scores = NULL;
        int32_t bits = 0, nodes = 0, graphs = 0;
        Py_BEGIN_ALLOW_THREADS
        slot = ((submit_fn)(uintptr_t)submit_addr)(h, ap, ai, w_fast ? aw : NULL, nn, (int32_t)n, isz ? isz : 4);
        if (slot >= 0) rc = ((result_fn)(uintptr_t)result_addr)(h, slot, &state, &totals, &rounds, &scores, &bits, &nodes, &graphs);
        Py_END_ALLOW_THREADS
        if (slot < 0 || rc != 0) {
            out = PyLong_FromLong(slot < 0 ? slot : rc);
            goto done;
        }
        PyObject* o_state = PyByteArray_FromStringAndSize((const char*)state, nodes);
        PyObject* o_totals = PyByteArray_FromStringAndSize((const char*)totals, (Py_ssize_t)graphs * 8);
        PyObject* o_rounds = PyByteArray_FromStringAndSize((const char*)rounds, (Py_ssize_t)graphs * 4);
        PyObject* o_scores = scores ? PyByteArray_FromStringAndSize((const char*)scores, (Py_ssize_t)nodes * 4) : (Py_INCREF(Py_None), Py_None);
        PyObject* o_nn = PyByteArray_FromStringAndSize((const char*)nn, (Py_ssize_t)n * 4);
        if (o_state && o_totals && o_rounds && o_scores && o_nn)
            out = Py_BuildValue("(NNNNiN)", o_state, o_totals, o_rounds, o_scores, (int)bits, o_nn);
        else {
            Py_XDECREF(o_state); Py_XDECREF(o_totals); Py_XDECREF(o_rounds); Py_XDECREF(o_scores); Py_XDECREF(o_nn);
        }
    }
done:
    for (int i = 0; i < held; ++i) PyBuffer_Release(&bufs[i]);
    Py_XDECREF(w_fast);
    Py_DECREF(ix_fast);
    Py_DECREF(ip_fast);
    return out;
}

static PyMethodDef methods[] = {{"addresses", addresses, METH_VARARGS, "fill a table of buffer addresses"},
                                {"solve_lists", solve_lists, METH_VARARGS, "dgcn_host_solver_submit + _result in one call"},
                                {NULL, NULL, 0, NULL}};
static struct PyModuleDef moddef = {PyModuleDef_HEAD_INIT, "_pyptr", "pointer-table helper of distgcn_amd", -1, methods};
PyMODINIT_FUNC PyInit__pyptr(void) { return PyModule_Create(&moddef); }
