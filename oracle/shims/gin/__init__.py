"""TEST INFRASTRUCTURE ONLY -- oracle shim for ``gin-config`` (reference
requirements.txt:3).  Tier A of SURVEY.md Appendix A: decorators keep classes
untouched and wrap plain functions / external configurables so that default
keyword bindings registered with :func:`bind` are applied (this is how the
bindings ``cc.Conv1d.bias = False`` (configs/v1.gin:33-34),
``blocks.normalization.mode = 'weight_norm'`` (v1.gin:41) and
``cc.get_padding.mode = 'causal'`` (causal.gin:5) are emulated).  All other
bindings of the .gin files are passed by the oracle builders
(oracle/ref_models.py) as explicit keyword arguments.  ``get_configurable``
raises ValueError for unknown names because rave/__init__.py:15-19 relies on it.
"""
import functools
import inspect

_REGISTRY = {}
_BINDINGS = {}
_SEARCH = []


def bind(name, **kwargs):
    _BINDINGS.setdefault(name, {}).update(kwargs)


def clear_bindings():
    _BINDINGS.clear()


def _wrap(fn, key):
    @functools.wraps(fn)
    def wrapper(*args, **kwargs):
        for k, v in _BINDINGS.get(key, {}).items():
            kwargs.setdefault(k, v)
        return fn(*args, **kwargs)

    return wrapper


def configurable(fn_or_name=None, **kwargs):
    def deco(fn):
        key = getattr(fn, "__name__", str(fn))
        if inspect.isclass(fn):
            # keep the class object (isinstance / inheritance), inject bindings through __init__
            orig = fn.__init__

            @functools.wraps(orig)
            def init(self, *args, **kwargs):
                if type(self) is fn or type(self).__init__ is init:
                    for k, v in _BINDINGS.get(key, {}).items():
                        kwargs.setdefault(k, v)
                return orig(self, *args, **kwargs)

            fn.__init__ = init
            out = fn
        else:
            out = _wrap(fn, key)
        _REGISTRY[key] = out
        return out

    if callable(fn_or_name):
        return deco(fn_or_name)
    return deco


def external_configurable(fn, name=None, module=None, **kwargs):
    key = f"{module}.{name or fn.__name__}" if module else (name or fn.__name__)
    out = _wrap(fn, key)
    _REGISTRY[key] = out
    return out


def get_configurable(name):
    if name not in _REGISTRY:
        raise ValueError(f"No configurable matching '{name}'.")
    return _REGISTRY[name]


def add_config_file_search_path(p):
    _SEARCH.append(str(p))


def operative_config_str():
    return ""


def clear_config(*a, **k):
    clear_bindings()


def enter_interactive_mode():
    pass


def parse_config_files_and_bindings(*a, **k):
    raise NotImplementedError("oracle gin shim: bindings are passed explicitly")


parse_config_file = parse_config_files_and_bindings


def bind_parameter(*a, **k):
    raise NotImplementedError


def get_bindings(*a, **k):
    return {}


# --------------------------------------------------------------------------------------------------
# Tier B: a parser for the syntax subset the reference's .gin files use, so that
# the REAL config files (configs/v1.gin, v2.gin, ...) plus an overlay (rave_amd/configs/mi355x.gin) can
# drive the construction of the unmodified ``rave.RAVE``:
#   from __gin__ import dynamic_registration / import a.b [as c] / from a import b / include "file.gin"
#   MACRO = value
#   [scope/]name.param = value            and the block form  "[scope/]name:" + indented "param = value"
#   values: Python literals, %MACRO, @[scope/]name (reference), @[scope/]name() (evaluated reference)
# Semantics kept: bindings are applied as keyword DEFAULTS when a configurable is called -- through a
# reference, or directly for objects decorated with @gin.configurable / external_configurable; scopes opened by a
# scoped reference stay active for nested calls; later bindings / macros override earlier ones.
# --------------------------------------------------------------------------------------------------
import ast as _ast
import importlib as _importlib
import os as _os
import re as _re

_MACROS = {}
_SCOPED = {}          # (scope, key) -> {param: value}
_ACTIVE_SCOPES = []
_NS = {}              # names imported by the parsed .gin files


class _Macro:
    def __init__(self, name):
        self.name = name


class _Ref:
    def __init__(self, spec, call):
        self.scope, _, self.name = spec.rpartition("/")
        self.call = call


def _key_of(obj):
    k = getattr(obj, "__gin_key__", None)
    if k is not None:
        return k
    return f"{getattr(obj, '__module__', '?')}.{getattr(obj, '__qualname__', repr(obj))}"


def _lookup(name):
    """dotted name -> Python object, through the .gin file's own imports (dynamic registration)."""
    parts = name.split(".")
    if parts[0] in _NS:
        obj = _NS[parts[0]]
        for p in parts[1:]:
            obj = getattr(obj, p)
        return obj
    if name in _REGISTRY:
        return _REGISTRY[name]
    raise ValueError(f"No configurable matching '{name}'.")


def _resolve(v):
    if isinstance(v, _Macro):
        if v.name not in _MACROS:
            raise ValueError(f"gin macro %{v.name} is not defined")
        return _resolve(_MACROS[v.name])
    if isinstance(v, _Ref):
        fn = _make_ref(_lookup(v.name), v.scope)
        return fn() if v.call else fn
    if isinstance(v, list):
        return [_resolve(x) for x in v]
    if isinstance(v, tuple):
        return tuple(_resolve(x) for x in v)
    if isinstance(v, dict):
        return {k: _resolve(x) for k, x in v.items()}
    return v


def _bound_kwargs(key):
    out = dict(_BINDINGS.get(key, {}))
    for s in _ACTIVE_SCOPES:
        out.update(_SCOPED.get((s, key), {}))
    return {k: _resolve(v) for k, v in out.items()}


def _make_ref(obj, scope):
    key = _key_of(obj)
    wrapped = hasattr(obj, "__gin_key__")          # decorated objects inject their own (unscoped + active-scope) bindings

    def call(*args, **kwargs):
        if scope:
            _ACTIVE_SCOPES.append(scope)
        try:
            if not wrapped:
                for k, v in _bound_kwargs(key).items():
                    kwargs.setdefault(k, v)
            return obj(*args, **kwargs)
        finally:
            if scope:
                _ACTIVE_SCOPES.pop()

    call.__gin_target__ = obj
    return call


# decorated configurables: extend the Tier-A wrappers so that they see parsed bindings (macros, references, scopes)
def _wrap(fn, key):      # noqa: F811  (replaces the Tier-A helper above)
    @functools.wraps(fn)
    def wrapper(*args, **kwargs):
        for k, v in _bound_kwargs(key).items():
            kwargs.setdefault(k, v)
        return fn(*args, **kwargs)

    wrapper.__gin_key__ = key
    return wrapper


_tier_a_configurable = configurable


def configurable(fn_or_name=None, **kwargs):      # noqa: F811
    def deco(fn):
        key = getattr(fn, "__name__", str(fn))
        if inspect.isclass(fn):
            orig = fn.__init__

            @functools.wraps(orig)
            def init(self, *args, **kw):
                if type(self) is fn or type(self).__init__ is init:
                    for k, v in _bound_kwargs(key).items():
                        kw.setdefault(k, v)
                return orig(self, *args, **kw)

            fn.__init__ = init
            fn.__gin_key__ = key
            out = fn
        else:
            out = _wrap(fn, key)
        _REGISTRY[key] = out
        return out

    if callable(fn_or_name):
        return deco(fn_or_name)
    return deco


def external_configurable(fn, name=None, module=None, **kwargs):      # noqa: F811
    key = f"{module}.{name or fn.__name__}" if module else (name or fn.__name__)
    out = _wrap(fn, key)
    _REGISTRY[key] = out
    return out


_TOKEN = _re.compile(r"@([\w/\.]+)(\(\))?|%(\w+)")


def _parse_value(text):
    def sub(m):
        if m.group(3):
            return f"__M__({m.group(3)!r})"
        return f"__R__({m.group(1)!r}, {bool(m.group(2))})"

    return eval(compile(_ast.parse(_TOKEN.sub(sub, text.strip()), mode="eval"), "<gin>", "eval"),
                {"__builtins__": {}}, {"__M__": _Macro, "__R__": _Ref, "True": True, "False": False, "None": None})


class _MissingModule:
    """A module a .gin file imports but this container cannot (rave.dataset needs udls / lmdb data files): bindings
    that target it are recorded in SKIPPED instead of failing the whole parse."""

    def __init__(self, name):
        self.__missing__ = name

    def __getattr__(self, item):
        raise ValueError(f"gin (shim): {self.__missing__} could not be imported here")


SKIPPED = []


def _bind_parsed(target, param, value):
    scope, _, name = target.rpartition("/")
    try:
        obj = _lookup(name)
    except ValueError as e:
        if "could not be imported" in str(e):
            SKIPPED.append(f"{target}.{param}")
            return
        raise
    key = _key_of(obj)
    if scope:
        _SCOPED.setdefault((scope, key), {})[param] = value
    else:
        _BINDINGS.setdefault(key, {})[param] = value


def _find(path):
    for base in [""] + _SEARCH:
        p = _os.path.join(base, path)
        if _os.path.isfile(p):
            return p
    raise IOError(f"gin: config file {path!r} not found in {_SEARCH}")


def _strip_comment(raw):
    quote = None
    for i, ch in enumerate(raw):
        if quote:
            if ch == quote:
                quote = None
        elif ch in "'\"":
            quote = ch
        elif ch == "#":
            return raw[:i]
    return raw


def _logical_lines(text):
    """Physical lines joined while brackets are open; comments stripped."""
    buf, depth = "", 0
    for raw in text.splitlines():
        line = _strip_comment(raw).rstrip()
        if not line.strip() and depth == 0:
            continue
        buf = (buf + " " + line.strip()) if buf else line
        depth += sum(line.count(c) for c in "([{") - sum(line.count(c) for c in ")]}")
        if depth <= 0:
            yield buf
            buf, depth = "", 0
    if buf:
        yield buf


def parse_config(text):      # noqa: C901
    block = None
    for line in _logical_lines(text):
        indented = line[:1] in " \t"
        s = line.strip()
        if not indented:
            block = None
        if s.startswith("from __gin__"):
            continue
        m = _re.match(r"import\s+([\w\.]+)(?:\s+as\s+(\w+))?$", s)
        if m:
            mod = _importlib.import_module(m.group(1))
            if m.group(2):
                _NS[m.group(2)] = mod
            else:
                _NS[m.group(1).split(".")[0]] = _importlib.import_module(m.group(1).split(".")[0])
            continue
        m = _re.match(r"from\s+([\w\.]+)\s+import\s+(\w+)(?:\s+as\s+(\w+))?$", s)
        if m:
            try:
                obj = _importlib.import_module(m.group(1) + "." + m.group(2))
            except ImportError:
                try:
                    obj = getattr(_importlib.import_module(m.group(1)), m.group(2))
                except (ImportError, AttributeError):
                    obj = _MissingModule(m.group(1) + "." + m.group(2))
            _NS[m.group(3) or m.group(2)] = obj
            continue
        m = _re.match(r"include\s+['\"](.+)['\"]$", s)
        if m:
            parse_config_file(m.group(1))
            continue
        m = _re.match(r"([\w/\.]+)\s*:$", s)
        if m and not indented:
            block = m.group(1)
            continue
        m = _re.match(r"([\w/\.]+)\s*=\s*(.+)$", s, _re.S)
        if not m:
            raise SyntaxError(f"gin (shim): cannot parse {line!r}")
        lhs, rhs = m.group(1), m.group(2)
        value = _parse_value(rhs)
        if indented and block:
            _bind_parsed(block, lhs, value)
        elif "." not in lhs and "/" not in lhs:
            _MACROS[lhs] = value
        else:
            target, _, param = lhs.rpartition(".")
            _bind_parsed(target, param, value)


def parse_config_file(path, *a, **k):      # noqa: F811
    with open(_find(path)) as f:
        parse_config(f.read())


def parse_config_files_and_bindings(config_files=None, bindings=None, *a, **k):      # noqa: F811
    for f in config_files or []:
        parse_config_file(f)
    for b in bindings or []:
        parse_config(b)


def bind_parameter(name, value):      # noqa: F811
    target, _, param = name.rpartition(".")
    try:
        _bind_parsed(target, param, value)
    except ValueError:
        _BINDINGS.setdefault(target.split(".")[-1], {})[param] = value


def clear_config(*a, **k):      # noqa: F811
    _BINDINGS.clear()
    _SCOPED.clear()
    _MACROS.clear()
    _NS.clear()
    del _ACTIVE_SCOPES[:]


def clear_bindings():      # noqa: F811
    _BINDINGS.clear()
    _SCOPED.clear()
