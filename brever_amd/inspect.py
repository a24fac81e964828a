"""Command-line option specs derived from type-hinted signatures.

Interface of the reference's brever/inspect.py:9-143 -- the contract that lets a model's
``__init__`` signature generate its ``config.yaml`` section and its CLI flags (README
"constructor args must be fully type-hinted with defaults"): ``get_func_spec(func)`` maps
every parsable argument to the ``argparse.add_argument`` keywords
``{type, action, default, required}``. Pinned against the reference's own specs for
BreverDataset, BreverTrainer and every in-scope model (tests/golden/config.json).

Typing rules (reference inspect.py:32-128):
* ``NoParse[T]`` arguments are skipped (objects that cannot come from a command line);
* ``list[T] / set[T] / tuple[T, ...]`` parse from one comma-separated string;
* a union needs exactly one parsable member: the one wrapped in ``Parse[...]``, else the one
  that is not ``NoParse[...]`` -- anything else is ambiguous and rejected;
* the leaf type must be str / int / float / bool / Path; bools parse ``true/yes/1`` etc.;
* a class marked ``_is_submodel`` inherits its parent's spec and overrides it.
"""
import argparse
import inspect
from types import UnionType
from typing import Generic, TypeVar, Union, get_args, get_origin

T = TypeVar('T')

_LEAF_TYPES = (str, int, float, bool)
_CONTAINERS = (list, set, tuple)


class NoParse(Generic[T]):
    """Marks an argument (or union member) that never comes from the command line."""


class Parse(Generic[T]):
    """Marks the union member the command line parses into."""


class Path:
    """String type of filesystem paths: forward slashes, no trailing slash."""

    def __new__(cls, s):
        return s.replace('\\', '/').rstrip('/')


class Bool:
    """String -> bool for flags written ``--flag=true``."""

    _true, _false = ('true', 'yes', '1'), ('false', 'no', '0')

    def __new__(cls, s):
        low = s.lower()
        if low in cls._true:
            return True
        if low in cls._false:
            return False
        raise argparse.ArgumentTypeError(f'expected bool value, got {s}')


class OriginAction:
    """``argparse`` action factory: a comma-separated string becomes ``origin(type_(x) ...)``
    (``--layers=1,2,3`` -> ``[1, 2, 3]``; the empty string -> empty container)."""

    def __init__(self, origin, type_):
        self.origin = origin
        self.type_ = type_

    def __call__(self, *args, **kwargs):
        origin, type_ = self.origin, self.type_

        class _Split(argparse.Action):
            def __call__(self, parser, namespace, values, option_string=None):
                items = [type_(s) for s in values.split(',') if s != '']
                setattr(namespace, self.dest, origin(items))

        return _Split(*args, **kwargs)


def _container_spec(name, hint, origin, default):
    """(leaf type, action) of ``list[T]`` / ``set[T]`` / ``tuple[T, ...]``."""
    inner = get_args(hint)
    if origin is tuple:
        if any(t != inner[0] for t in inner):
            raise ValueError(f'unsupported typing for argument {name}, got {hint}')
        if default is not None and len(default) != len(inner):
            raise ValueError(f'default value of argument {name} does not match '
                             f'typing, got {default} and {hint}')
    elif len(inner) != 1:
        raise ValueError(f'unsupported typing for argument {name}, got {hint}')
    if default is not None and not (isinstance(default, origin)
                                    and all(isinstance(d, inner[0]) for d in default)):
        raise ValueError(f'default value of argument {name} does not match '
                         f'typing, got {default} and {hint}')
    return str, OriginAction(origin, inner[0])


def _union_member(name, hint):
    """The one member of a union the command line parses into."""
    marked = [t for t in get_args(hint) if get_origin(t) is Parse]
    plain = [t for t in get_args(hint)
             if get_origin(t) is not Parse and get_origin(t) is not NoParse]
    ambiguous = ValueError(f'ambiguous union typing for argument {name}, got {hint}; '
                           'use Parse or NoParse to avoid ambiguity')
    if len(marked) > 1:
        raise ambiguous
    if marked:
        return get_args(marked[0])[0]
    if len(plain) > 1:
        raise ambiguous
    if not plain:
        raise ValueError(f'unsupported typing for argument {name}, got {hint}')
    # (the reference unwraps the single remaining member with get_args: a bare type there
    # has no args and fails; every in-scope signature uses Parse[...] in its unions)
    inner = get_args(plain[0])
    return inner[0] if inner else plain[0]


def get_func_spec(func):
    """``{argument: dict(type, action, default, required)}`` for every parsable argument of
    ``func`` (a function, or a class for its ``__init__``)."""
    target = func.__init__ if inspect.isclass(func) else func
    sig = inspect.signature(target)
    hints = getattr(target, '__annotations__', {})
    spec = {}
    for name, prm in sig.parameters.items():
        if name == 'self' or prm.kind in (prm.VAR_POSITIONAL, prm.VAR_KEYWORD):
            continue
        if name not in hints:
            raise ValueError(f'missing type hint for argument {name}')
        hint = hints[name]
        has_default = prm.default is not inspect.Parameter.empty
        default = prm.default if has_default else None
        action = None
        origin = get_origin(hint)
        if origin is NoParse:
            continue
        if origin in _CONTAINERS:
            leaf, action = _container_spec(name, hint, origin, default)
        elif origin in (Union, UnionType):
            leaf = _union_member(name, hint)
        elif origin is None:
            leaf = hint
        else:
            raise ValueError(f'unsupported typing for argument {name}, got {hint}')
        if action is None and default is not None and leaf is not Path \
                and not isinstance(default, leaf):
            raise ValueError(f'default value of argument {name} does not match '
                             f'typing, got {default} and {hint}')
        if leaf not in _LEAF_TYPES and leaf is not Path:
            raise ValueError(f'unsupported typing for argument {name}, got {hint}')
        spec[name] = dict(type=Bool if leaf is bool else leaf, action=action,
                          default=default, required=not has_default)
    if getattr(func, '_is_submodel', False):
        merged = get_func_spec(func.__bases__[0])
        merged.update(spec)
        # sub-models written with **kwargs declare their overridden defaults in `_defaults`
        for name, value in func.__dict__.get('_defaults', {}).items():
            if name in merged:
                merged[name] = dict(merged[name], default=value, required=False)
        spec = merged
    return spec
