"""The launcher for scripts written against the reference's embedded `volpy` module (volren_amd/run_script.py; src/main.cpp:83-91,311-357) --
the parts that need no GPU: argument handling, `import volpy`, `__main__` / `__file__`, and the hand-over from the `volren` executable."""
import json
import os
import subprocess
import sys

import pytest

from volren_amd import run_script

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = """
import json, os, sys
import volpy
assert __name__ == "__main__"
v = volpy.vec3(3, 0, 4)
json.dump(dict(file=os.path.basename(__file__), argv=sys.argv[1:], length=v.length(), context=dict(volpy._CONTEXT),
               same=sys.modules["volpy"].__name__), open(os.environ["OUT_JSON"], "w"))
"""


def test_parse_takes_the_context_flags_and_leaves_the_rest():
    assert run_script.parse(["a.py", "--render", "-w", "640", "-h", "480"]) == ("a.py", 640, 480, None, [])
    assert run_script.parse(["-h", "48", "--title", "x", "--device", "2", "dir/b.py", "c.py", "--flag", "---debug"]) == ("dir/b.py", None, 48, 2, ["c.py", "--flag"])
    with pytest.raises(SystemExit):
        run_script.parse(["a.py", "-w"])


@pytest.mark.parametrize("launcher", ["module", "volren"])
def test_script_sees_volpy_and_the_context(tmp_path, launcher):
    script = tmp_path / "gen.py"
    script.write_text(SCRIPT)
    out_json = tmp_path / "out.json"
    env = dict(os.environ, OUT_JSON=str(out_json), PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    if launcher == "module":
        cmd = [sys.executable, "-m", "volren_amd.run_script", str(script), "-w", "96", "-h", "64", "--render", "it's"]
    else:
        exe = os.path.join(ROOT, "volren_amd", "volren")
        if not os.path.exists(exe):
            pytest.skip("volren is not built")
        cmd = [exe, str(script), "--render", "-w", "96", "-h", "64", "it's"]
        env.pop("PYTHONPATH")                                   # the executable finds the package next to itself
    out = subprocess.run(cmd, cwd=tmp_path, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    j = json.load(open(out_json))
    assert j["file"] == "gen.py" and j["argv"] == ["it's"] and abs(j["length"] - 5.0) < 1e-6
    assert j["context"]["width"] == 96 and j["context"]["height"] == 64 and j["same"] == "volren_amd.volpy"


def test_failing_script_is_reported(tmp_path):
    script = tmp_path / "bad.py"
    script.write_text("import volpy\nraise ValueError('boom')\n")
    out = subprocess.run([sys.executable, "-m", "volren_amd.run_script", str(script)], cwd=tmp_path, capture_output=True, text=True, timeout=300,
                         env=dict(os.environ, PYTHONPATH=ROOT))
    assert out.returncode != 0 and "boom" in out.stderr and "Error executing python script" in out.stderr
    none = subprocess.run([sys.executable, "-m", "volren_amd.run_script", "missing.py"], cwd=tmp_path, capture_output=True, text=True, timeout=300,
                          env=dict(os.environ, PYTHONPATH=ROOT))
    assert none.returncode != 0 and "no such script" in none.stderr
