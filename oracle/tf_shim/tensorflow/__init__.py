"""A NumPy stand-in for the slice of TensorFlow (tf.compat.v1 graph mode) that the reference's model code calls.
TEST INFRASTRUCTURE ONLY - used by oracle/run_reference.py in the build container, never by the product.

Why it exists.  The reference's GCN forward executes inside TensorFlow (`gcn/layers.py`, `gcn/models.py`,
`mwis_dqn_call.py`, `mwis_gdpg_call.py` all `import tensorflow`), TensorFlow is not installable here, and so
those modules could not even be imported: their behaviour (which layer gets which activation, bias or not, how the
supports are sliced, the variable names a checkpoint is restored by, zero-weight pruning and id mapping in
`solve_mwis`, the iterative solvers' control flow ...) was only RESTATED in oracle/ref_numpy.py.  With this package
first on sys.path the reference's own Python runs unmodified: it builds its graph out of the lazy nodes below and
`Session.run` evaluates them with NumPy in float32.  What that pins: everything the reference's Python decides.
What it does NOT pin: TensorFlow's kernels - `sparse_tensor_dense_matmul`, `matmul`, `add_n` are evaluated here
with SciPy / NumPy float32 arithmetic (the same calls oracle/ref_numpy.py makes), so the summation order inside a
TF kernel stays "parity unpinned" and is stated as such wherever these vectors are used.

Only what the inference path evaluates has numerics; the loss / optimiser sub-graphs are built as inert nodes
(evaluating one raises).  This is not TensorFlow and makes no attempt to be.
"""
import os
import sys
import types

import numpy as np
import scipy.sparse as _sp

float32, float64, int32, int64 = np.float32, np.float64, np.int32, np.int64
bool = np.bool_  # noqa: A001  (tf.bool)

_RNG = np.random.RandomState(20230600)


# ------------------------------------------------------------------------------------------------ scopes / names
_scope = []
_used_names = {}
_variables = []


class _Scope:
    def __init__(self, name):
        self.name = name

    def __enter__(self):
        _scope.append(self.name)
        return "/".join(_scope) + "/"

    def __exit__(self, *a):
        _scope.pop()


def name_scope(name, *a, **k):
    return _Scope(name)


def _unique(full):
    n = _used_names.get(full, 0)
    _used_names[full] = n + 1
    return full if n == 0 else "%s_%d" % (full, n)


# ------------------------------------------------------------------------------------------------ lazy nodes
class _Shape:
    def __init__(self, dims):
        self.dims = None if dims is None else list(dims)

    def as_list(self):
        return list(self.dims)

    def __getitem__(self, i):
        return self.dims[i]

    def __len__(self):
        return len(self.dims)


class SparseValue:
    """Evaluated sparse tensor (float32 values, row/col indices)."""

    def __init__(self, indices, values, shape):
        self.indices = np.asarray(indices).reshape(-1, 2)
        self.values = np.asarray(values)
        self.shape = tuple(int(s) for s in shape)

    def csr(self):
        return _sp.csr_matrix((self.values, (self.indices[:, 0], self.indices[:, 1])), shape=self.shape)


class Tensor:
    def __init__(self, fn=None, shape=None, name=None, what="node"):
        self._fn, self._shape, self.name, self._what = fn, _Shape(shape), name, what

    # --- evaluation
    def _eval(self, env):
        if id(self) in env["feed"]:
            return env["feed"][id(self)]
        if id(self) not in env["memo"]:
            if self._fn is None:
                raise NotImplementedError("tf_shim: '%s' is an inert node (only the inference path has numerics)" % self._what)
            env["memo"][id(self)] = self._fn(env)
        return env["memo"][id(self)]

    # --- static shape
    def get_shape(self):
        return self._shape

    @property
    def shape(self):
        return self._shape

    # --- arithmetic (float32 semantics come from the operands' dtypes)
    def _bin(self, other, f, what):
        o = other
        return Tensor(lambda env: f(_ev(self, env), _ev(o, env)), name=what, what=what)

    def __add__(self, o): return self._bin(o, lambda a, b: a + b, "add")
    def __radd__(self, o): return self._bin(o, lambda a, b: b + a, "add")
    def __sub__(self, o): return self._bin(o, lambda a, b: a - b, "sub")
    def __rsub__(self, o): return self._bin(o, lambda a, b: b - a, "sub")
    def __mul__(self, o): return self._bin(o, _mul, "mul")
    def __rmul__(self, o): return self._bin(o, lambda a, b: _mul(b, a), "mul")
    def __truediv__(self, o): return self._bin(o, lambda a, b: a / b, "div")
    def __rtruediv__(self, o): return self._bin(o, lambda a, b: b / a, "div")
    def __pow__(self, o): return self._bin(o, lambda a, b: a ** b, "pow")
    def __neg__(self): return Tensor(lambda env: -_ev(self, env), what="neg")

    def __getitem__(self, idx):
        return Tensor(lambda env: _ev(self, env)[idx], what="slice")

    def __hash__(self):
        return id(self)

    def __eq__(self, other):
        return self is other


class SparseTensor(Tensor):
    def __mul__(self, o):
        def f(env):
            s, k = _ev(self, env), _ev(o, env)
            return SparseValue(s.indices, s.values * np.float32(k), s.shape)
        return SparseTensor(f, shape=self._shape.dims, what="sparse_mul")


def _ev(x, env):
    if isinstance(x, Tensor):
        return x._eval(env)
    if isinstance(x, (list, tuple)) and any(isinstance(i, Tensor) for i in x):
        return [_ev(i, env) for i in x]
    return x


def _mul(a, b):
    if isinstance(a, SparseValue):
        return SparseValue(a.indices, a.values * np.float32(b), a.shape)
    return a * b


def _f32(x):
    a = np.asarray(x)
    return a.astype(np.float32) if a.dtype.kind == "f" else a


def _inert(what):
    def make(*a, **k):
        return Tensor(None, what=what)
    return make


# ------------------------------------------------------------------------------------------------ constants & ops
def constant(value, dtype=None, shape=None, name=None):
    arr = np.asarray(value, dtype=dtype)
    return Tensor(lambda env: arr, shape=arr.shape, what="constant")


def zeros(shape, dtype=float32, name=None):
    return Tensor(lambda env: np.zeros([int(s) for s in shape], dtype=dtype), shape=shape, what="zeros")


def ones(shape, dtype=float32, name=None):
    return Tensor(lambda env: np.ones([int(s) for s in shape], dtype=dtype), shape=shape, what="ones")


def eye(n, dtype=float32):
    return Tensor(lambda env: np.eye(int(n), dtype=dtype), shape=(n, n), what="eye")


def ones_like(x, name=None):
    return Tensor(lambda env: np.ones_like(_ev(x, env)), what="ones_like")


def cast(x, dtype=None, name=None):
    def f(env):
        v = _ev(x, env)
        if isinstance(v, SparseValue):
            return SparseValue(v.indices, v.values.astype(dtype), v.shape)
        return np.asarray(v).astype(dtype)
    cls = SparseTensor if isinstance(x, SparseTensor) else Tensor
    return cls(f, shape=getattr(x, "_shape", _Shape(None)).dims, what="cast")


def floor(x, name=None):
    return Tensor(lambda env: np.floor(_ev(x, env)), what="floor")


def matmul(a, b, name=None):
    return Tensor(lambda env: np.matmul(_f32(_ev(a, env)), _f32(_ev(b, env))), what="matmul")


def add_n(inputs, name=None):
    def f(env):
        vals = [_ev(t, env) for t in inputs]
        out = vals[0]
        for v in vals[1:]:
            out = out + v  # left to right, as Eigen's add_n does for a handful of inputs
        return out
    return Tensor(f, what="add_n")


def concat(values, axis=0, name=None):
    return Tensor(lambda env: np.concatenate([_ev(v, env) for v in values], axis=axis), what="concat")


def reshape(x, shape, name=None):
    return Tensor(lambda env: np.reshape(_ev(x, env), shape), what="reshape")


def argmax(x, axis=None, name=None, **k):
    return Tensor(lambda env: np.argmax(_ev(x, env), axis=0 if axis is None else axis).astype(np.int64), what="argmax")


def reduce_mean(x, axis=None, name=None, **k):
    def f(env):
        v = _ev(x, env)
        return np.mean(v, axis=axis, dtype=v.dtype if hasattr(v, "dtype") and v.dtype.kind == "f" else None)
    return Tensor(f, what="reduce_mean")


reduce_max = lambda x, axis=None, **k: Tensor(lambda env: np.max(_ev(x, env), axis=axis), what="reduce_max")  # noqa: E731
reduce_min = lambda x, axis=None, **k: Tensor(lambda env: np.min(_ev(x, env), axis=axis), what="reduce_min")  # noqa: E731
sqrt = lambda x, **k: Tensor(lambda env: np.sqrt(_ev(x, env)), what="sqrt")  # noqa: E731
square = lambda x, **k: Tensor(lambda env: np.square(_ev(x, env)), what="square")  # noqa: E731
abs = lambda x, **k: Tensor(lambda env: np.abs(_ev(x, env)), what="abs")  # noqa: E731,A001
equal = _inert("equal")
logical_and = _inert("logical_and")
logical_not = _inert("logical_not")
clip_by_value = _inert("clip_by_value")
norm = _inert("norm")


def zeros_initializer(*a, **k):
    return lambda shape, dtype=float32: np.zeros(shape, dtype=dtype)


def constant_initializer(value):
    return lambda shape, dtype=float32: np.asarray(value, dtype=dtype).reshape(shape)


# ------------------------------------------------------------------------------------------------ variables
class Variable(Tensor):
    def __init__(self, initial_value=None, name=None, trainable=True, dtype=None, **k):
        full = _unique("/".join(_scope + [name or "Variable"]))
        super().__init__(None, name=full + ":0", what="variable")
        self._initial = initial_value
        self.value_ = None
        self.trainable = trainable
        self._fn = self._read
        _variables.append(self)
        if isinstance(initial_value, Tensor):
            self._shape = initial_value._shape

    def _read(self, env):
        if self.value_ is None:
            raise RuntimeError("tf_shim: variable %s read before initialisation" % self.name)
        return self.value_

    def initialize(self):
        v = self._initial
        if isinstance(v, Tensor):
            v = v._eval({"feed": {}, "memo": {}})
        self.value_ = np.array(v)

    def _eval(self, env):  # variables are never memoised: their value may change between runs
        return self._read(env)


# ------------------------------------------------------------------------------------------------ sub-modules
class _NS(types.SimpleNamespace):
    pass


def _leaky_relu(x, alpha=0.2, name=None):
    def f(env):
        v = _ev(x, env)
        return np.where(v > 0, v, v.dtype.type(alpha) * v)
    return Tensor(f, what="leaky_relu")


def _relu(x, name=None):
    return Tensor(lambda env: np.maximum(_ev(x, env), 0), what="relu")


def _dropout(x, rate=0.0, keep_prob=None, name=None, **k):
    def f(env):
        r = float(_ev(rate, env)) if keep_prob is None else 1.0 - float(_ev(keep_prob, env))
        if r != 0.0:
            raise NotImplementedError("tf_shim: dropout with rate %g (inference feeds 0)" % r)
        return _ev(x, env)
    return Tensor(f, what="dropout")


def _softmax(x, axis=-1, name=None):
    def f(env):
        v = _ev(x, env)
        e = np.exp(v - v.max(axis=axis, keepdims=True))
        return e / e.sum(axis=axis, keepdims=True)
    return Tensor(f, what="softmax")


nn = _NS(leaky_relu=_leaky_relu, relu=_relu, dropout=_dropout, softmax=_softmax, l2_loss=_inert("l2_loss"),
         softmax_cross_entropy_with_logits=_inert("xent"), weighted_cross_entropy_with_logits=_inert("wxent"))
math = _NS(reduce_std=_inert("reduce_std"), multiply=lambda a, b, **k: a * b)
losses = _NS(mean_squared_error=_inert("mse"))
summary = _NS(histogram=lambda *a, **k: None, scalar=lambda *a, **k: None, create_file_writer=lambda *a, **k: None)


class _CkptState:
    def __init__(self, path):
        self.model_checkpoint_path = path


def _get_checkpoint_state(name):
    idx = os.path.join(name, "model.ckpt.index")
    return _CkptState(os.path.join(name, "model.ckpt")) if os.path.isfile(idx) else None


class _Adam:
    def __init__(self, *a, **k):
        pass

    def minimize(self, *a, **k):
        return Tensor(None, what="opt_op")

    def compute_gradients(self, *a, **k):
        return []


train = _NS(get_checkpoint_state=_get_checkpoint_state, AdamOptimizer=_Adam)
layers = _NS()


# ------------------------------------------------------------------------------------------------ flags (absl style)
class _Flags:
    def __init__(self):
        object.__setattr__(self, "_vals", {})

    def __getattr__(self, k):
        try:
            return object.__getattribute__(self, "_vals")[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self._vals[k] = v

    def __contains__(self, k):
        return k in self._vals


class _FlagsModule:
    FLAGS = _Flags()

    @classmethod
    def _define(cls, conv, name, default, help=""):
        val = default
        for a in sys.argv[1:]:  # --name=value on the command line wins, as with absl
            if a.startswith("--" + name + "="):
                raw = a.split("=", 1)[1]
                val = (raw.lower() in ("1", "true", "yes")) if conv is bool_conv else conv(raw)
        cls.FLAGS._vals[name] = val

    @classmethod
    def DEFINE_string(cls, n, d, h=""): cls._define(str, n, d, h)
    @classmethod
    def DEFINE_float(cls, n, d, h=""): cls._define(float, n, d, h)
    @classmethod
    def DEFINE_integer(cls, n, d, h=""): cls._define(int, n, d, h)
    @classmethod
    def DEFINE_bool(cls, n, d, h=""): cls._define(bool_conv, n, d, h)


def bool_conv(x):
    return x


# ------------------------------------------------------------------------------------------------ tf.compat.v1
def _placeholder(dtype=None, shape=None, name=None):
    return Tensor(None, shape=shape, name=name, what="placeholder")


def _sparse_placeholder(dtype=None, shape=None, name=None):
    t = SparseTensor(None, shape=shape, name=name, what="sparse_placeholder")
    t._dtype = dtype
    return t


def _placeholder_with_default(value, shape=None, name=None):
    return Tensor(lambda env: np.float32(value) if isinstance(value, float) else np.asarray(value), shape=shape,
                  what="placeholder_with_default")


def _random_uniform(shape, minval=0, maxval=1, dtype=float32, seed=None, name=None):
    def f(env):
        s = _ev(shape, env)
        s = [int(i) for i in (s if np.ndim(s) else [s])]
        return _RNG.uniform(minval, maxval, size=s).astype(dtype)
    return Tensor(f, shape=shape if not isinstance(shape, Tensor) else None, what="random_uniform")


def _sparse_tensor_dense_matmul(a, b, name=None, **k):
    def f(env):
        s, d = _ev(a, env), _f32(_ev(b, env))
        return np.asarray(s.csr().astype(np.float32).dot(d), dtype=np.float32)
    return Tensor(f, what="sparse_tensor_dense_matmul")


def _sparse_retain(sp_input, to_retain):
    def f(env):
        s, m = _ev(sp_input, env), np.asarray(_ev(to_retain, env)).astype(np.bool_)
        return SparseValue(s.indices[m], s.values[m], s.shape)
    return SparseTensor(f, shape=sp_input._shape.dims, what="sparse_retain")


def _get_variable(name, shape=None, initializer=None, trainable=True, dtype=float32, **k):
    init = initializer() if isinstance(initializer, type(zeros_initializer)) and initializer is zeros_initializer else initializer
    val = init(shape if shape is not None else [], dtype) if callable(init) else init
    return Variable(val, name=name, trainable=trainable)


def _global_variables_initializer():
    def f(env):
        for v in _variables:
            if v.value_ is None:
                v.initialize()
        return None
    return Tensor(f, what="init")


def _dense(inputs, units, kernel_initializer=None, use_bias=True, name=None, **k):
    with _Scope(name or "dense"):
        in_dim = inputs._shape.dims[-1] if inputs._shape.dims else None
        if in_dim is None:
            raise NotImplementedError("tf_shim: layers.dense needs a static input width")
        lim = np.sqrt(6.0 / (in_dim + int(units)))
        init = kernel_initializer([in_dim, int(units)], float32) if callable(kernel_initializer) else \
            _RNG.uniform(-lim, lim, size=(in_dim, int(units))).astype(np.float32)
        kernel = Variable(init, name="kernel")
        bias = Variable(np.zeros(int(units), np.float32), name="bias")
    out = matmul(inputs, kernel) + bias
    out._shape = _Shape([None, int(units)])
    return out


class _Saver:
    def __init__(self, *a, **k):
        pass

    def restore(self, sess, path):
        """Assign every global variable from the bundle at ``path`` by its name (TF: Saver.restore)."""
        from distgcn_amd.checkpoint import load_bundle  # the build's own TF-V2 bundle reader (crc-checked)
        tensors = load_bundle(path)
        for v in _variables:
            key = v.name[:-2] if v.name.endswith(":0") else v.name
            if key not in tensors:
                raise KeyError("tf_shim Saver.restore: %s not in checkpoint %s" % (key, path))
            if v.value_ is not None and tuple(np.shape(v.value_)) != tuple(tensors[key].shape):
                raise ValueError("tf_shim Saver.restore: shape mismatch for %s: %s vs %s" % (key, np.shape(v.value_), tensors[key].shape))
            v.value_ = np.array(tensors[key])

    def save(self, sess, path):
        from distgcn_amd.checkpoint import save_bundle
        save_bundle(path, {(v.name[:-2] if v.name.endswith(":0") else v.name): v.value_ for v in _variables})


class _Session:
    def __init__(self, *a, **k):
        pass

    def as_default(self):
        return self

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False

    def run(self, fetches, feed_dict=None):
        feed = {}
        for k, v in (feed_dict or {}).items():
            if isinstance(k, SparseTensor):
                idx, vals, shape = v
                feed[id(k)] = SparseValue(idx, np.asarray(vals).astype(np.float32), shape)  # sparse placeholders are tf.float32
            else:
                feed[id(k)] = v
        env = {"feed": feed, "memo": {}}
        if isinstance(fetches, (list, tuple)):
            return [f._eval(env) for f in fetches]
        return fetches._eval(env)

    def close(self):
        pass


class _ConfigProto:
    def __init__(self, *a, **k):
        self.gpu_options = types.SimpleNamespace(allow_growth=False)


class _GraphKeys:
    GLOBAL_VARIABLES = "variables"
    TRAINABLE_VARIABLES = "trainable_variables"


_v1_layers = _NS(dense=_dense)
layers.dense = _dense
_v1_train = _NS(Saver=_Saver, AdamOptimizer=_Adam, exponential_decay=_inert("exponential_decay"))
compat = _NS(v1=_NS(
    placeholder=_placeholder, sparse_placeholder=_sparse_placeholder, placeholder_with_default=_placeholder_with_default,
    random_uniform=_random_uniform, sparse_tensor_dense_matmul=_sparse_tensor_dense_matmul, sparse_retain=_sparse_retain,
    variable_scope=lambda name, *a, **k: _Scope(name), get_variable=_get_variable,
    get_collection=lambda key, scope=None: [v for v in _variables if scope is None or v.name.startswith(scope)],
    trainable_variables=lambda: [v for v in _variables if v.trainable],
    global_variables_initializer=_global_variables_initializer, GraphKeys=_GraphKeys, Session=_Session,
    ConfigProto=_ConfigProto, disable_eager_execution=lambda: None, flags=_FlagsModule, train=_v1_train, layers=_v1_layers,
    diag=_inert("diag"), metrics=_NS(root_mean_squared_error=_inert("rmse")), assign=_inert("assign"), group=_inert("group")))
app = _NS(flags=_FlagsModule)
random = _NS(uniform=_random_uniform)
__version__ = "0.0-numpy-shim"


def reset_shim_state(seed=20230600):
    """Forget every variable / name (a new 'graph') and reseed the initialiser stream."""
    global _RNG
    _variables.clear()
    _used_names.clear()
    del _scope[:]
    _RNG = np.random.RandomState(seed)


def shim_variables():
    return {(v.name[:-2] if v.name.endswith(":0") else v.name): v.value_ for v in _variables}
