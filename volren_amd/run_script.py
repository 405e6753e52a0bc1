"""Runs a script written for the reference's embedded Python module, unmodified:

    python -m volren_amd.run_script <script.py> [-w W] [-h H] [--device D] [--render] [script arguments ...]

The reference starts such scripts as `./volren script.py --render -w W -h H` (src/main.cpp:83-91: pybind11::eval_file inside the
executable, whose embedded module `volpy` -- src/bindings.cpp:64-209 -- exists nowhere else).  Here the script runs in an ordinary
interpreter: `import volpy` finds volren_amd.volpy (registered in sys.modules before the script starts), `volpy.Renderer()` gets the
resolution of -w / -h like the reference's renderer gets the GL context's (src/main.cpp:311-357, src/renderer.cpp:47), and the file is
executed as `__main__` with `__file__` set, so `if __name__ == "__main__":` blocks and `os.path.dirname(__file__)` work as they do
under eval_file.  The window flags of the reference's command line are accepted and ignored; `--render` needs nothing (there is no window).
`volren script.py ...` (volren_amd/csrc/main.cpp) hands over to this module.
"""
import os
import runpy
import sys

_IGNORED_WITH_VALUE = ("--title", "--major", "--minor", "--swap", "--font", "--fontsize")
_IGNORED = ("--render", "--no-resize", "--hidden", "--no-decoration", "--floating", "--maximised", "---debug")


def parse(argv):
    """(script, width, height, device, rest): the first *.py argument is the script; -w / -h / --device as in src/main.cpp:311-357."""
    script, width, height, device, rest = None, None, None, None, []
    i = 0
    while i < len(argv):
        a = argv[i]
        if a in ("-w", "-h", "--device") or a in _IGNORED_WITH_VALUE:
            if i + 1 >= len(argv):
                raise SystemExit("run_script: missing value after %s" % a)
            v = argv[i + 1]
            i += 1
            if a == "-w":
                width = int(v)
            elif a == "-h":
                height = int(v)
            elif a == "--device":
                device = int(v)
        elif a in _IGNORED:
            pass
        elif script is None and a.endswith(".py"):
            script = a
        else:
            rest.append(a)
        i += 1
    return script, width, height, device, rest


def run(script, width=None, height=None, device=None, args=()):
    """Execute `script` as __main__ with `volpy` importable; returns the script's globals."""
    from . import volpy
    volpy.set_context(width, height, device)
    sys.modules["volpy"] = volpy
    old_argv = sys.argv
    sys.argv = [script] + list(args)
    try:
        return runpy.run_path(script, run_name="__main__")
    finally:
        sys.argv = old_argv


def main(argv=None):
    script, width, height, device, rest = parse(list(sys.argv[1:] if argv is None else argv))
    if script is None:
        raise SystemExit(__doc__)
    if not os.path.isfile(script):
        raise SystemExit("run_script: no such script: %s" % script)
    try:
        run(script, width, height, device, rest)
    except SystemExit:
        raise
    except Exception as e:                                 # the reference prints and goes on (src/main.cpp:88-90); a launcher has nothing to go on with
        import traceback
        traceback.print_exc()
        raise SystemExit("Error executing python script %s: %s" % (script, e))


if __name__ == "__main__":
    main()
