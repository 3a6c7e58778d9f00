"""A minimal stand-in for the part of Sacred the reference's train.py uses (train.py:7-10,26-63,246,
408,484): ``Experiment`` with ``config`` / ``capture`` / ``command`` / ``automain`` decorators,
``run_commandline`` parsing ``[command] with key=value ...`` (values are Python literals, bare words
stay strings), and a run object with ``log_scalar`` and ``_id``.  Used only when the real ``sacred``
package is not installed; the CLI surface is the same either way.
"""
import ast
import inspect
import logging
import sys


class Run:
    """What a command sees as ``_run``: scalar log (also kept in ``.scalars``) and an id."""

    def __init__(self, config, run_id=None):
        self.config = config
        self._id = run_id  # None without an observer, like Sacred (files are then named model-None.pt)
        self.scalars = {}
        self.info = {}

    def log_scalar(self, name, value, step=None):
        self.scalars.setdefault(name, []).append((step, float(value)))


def _locals_of(fn):
    """Run a config function and return its local variables (Sacred's config-scope semantics)."""
    captured = {}

    def profiler(frame, event, arg):
        if event == "return" and frame.f_code is fn.__code__:
            captured.update(frame.f_locals)

    old = sys.getprofile()
    sys.setprofile(profiler)
    try:
        fn()
    finally:
        sys.setprofile(old)
    return {k: v for k, v in captured.items() if not k.startswith("_")}


def _parse_value(text):
    try:
        return ast.literal_eval(text)
    except (ValueError, SyntaxError):
        return text


class Experiment:
    def __init__(self, name=None, **kwargs):
        self.name = name
        self.observers = []
        self.logger = None
        self._config_fns = []
        self._commands = {}
        self._default_command = None
        self.current_run = None

    # ------------------------------------------------------------------ decorators
    def config(self, fn):
        self._config_fns.append(fn)
        return fn

    def _inject(self, fn):
        signature = inspect.signature(fn)

        def wrapper(*args, **kwargs):
            run = self.current_run
            bound = signature.bind_partial(*args, **kwargs)
            for name in signature.parameters:
                if name in bound.arguments:
                    continue
                if name == "_run":
                    kwargs[name] = run
                elif name == "_log":
                    kwargs[name] = self.logger or logging.getLogger(fn.__name__)
                elif name == "_config":
                    kwargs[name] = dict(run.config) if run else {}
                elif run is not None and name in run.config:
                    kwargs[name] = run.config[name]
            return fn(*args, **kwargs)

        wrapper.__name__ = fn.__name__
        wrapper.__doc__ = fn.__doc__
        wrapper.__wrapped__ = fn
        return wrapper

    def capture(self, fn):
        return self._inject(fn)

    def command(self, fn):
        wrapped = self._inject(fn)
        self._commands[fn.__name__] = wrapped
        return wrapped

    def main(self, fn):
        wrapped = self.command(fn)
        self._default_command = fn.__name__
        return wrapped

    def automain(self, fn):
        wrapped = self.main(fn)
        if fn.__module__ == "__main__":
            self.run_commandline()
        return wrapped

    # ------------------------------------------------------------------ running
    def build_config(self, updates=None):
        config = {}
        for fn in self._config_fns:
            config.update(_locals_of(fn))
        unknown = [k for k in (updates or {}) if k not in config]
        if unknown and config:
            raise KeyError(f"unknown config keys {unknown}; known: {sorted(config)}")
        config.update(updates or {})
        return config

    def run(self, command_name=None, config_updates=None):
        name = command_name or self._default_command
        if name not in self._commands:
            raise KeyError(f"unknown command {name!r}; available: {sorted(self._commands)}")
        run = Run(self.build_config(config_updates))
        self.current_run = run
        run.result = self._commands[name]()
        return run

    def run_commandline(self, argv=None):
        argv = list(sys.argv if argv is None else argv)[1:]
        command, updates = None, {}
        if argv and argv[0] != "with":
            command = argv.pop(0)
        if argv and argv[0] == "with":
            for item in argv[1:]:
                key, sep, value = item.partition("=")
                if not sep:
                    raise ValueError(f"expected key=value after 'with', got {item!r}")
                updates[key] = _parse_value(value)
        return self.run(command, updates)
