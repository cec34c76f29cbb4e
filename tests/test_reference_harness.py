"""the reference's own regression scripts (tools/tests/test_*.py + helperInclude.py) run UNMODIFIED against the
package: first with MANTA_GEN_TEST_DATA=1 (writes <tmp>/testdata/*.uni through Grid.save), then in check mode (Grid.load +
gridMaxDiff* + the script's own thresholds -> "OK!" lines).  Same backend in both passes (the oracle library), so this pins the
harness surface -- introspection attributes, helper plugins, .uni round trip, every plugin signature the scripts use -- not the
numerics (those are pinned against the compiled reference elsewhere).  The scripts are read from /root/reference at test time."""
import contextlib
import io
import os
import shutil
import sys

import pytest

TD = "/root/reference/tools/tests"
pytestmark = pytest.mark.skipif(not os.path.isdir(TD), reason="reference tests not present on this machine")


def run_script(name, workdir, gen):
    src = open(os.path.join(TD, name)).read()
    path = os.path.join(workdir, "tests", name)
    old_argv, old_cwd, old_path = sys.argv, os.getcwd(), list(sys.path)
    sys.argv = [path]
    sys.path.insert(0, TD)                    # helperInclude / helperGeneric
    os.environ["MANTA_GEN_TEST_DATA"] = "1" if gen else "0"
    os.chdir(os.path.join(workdir, "tests"))
    buf = io.StringIO()
    try:
        with contextlib.redirect_stdout(buf):
            exec(compile(src, name, "exec"), {"__name__": "__main__", "__file__": path})
    finally:
        sys.argv, sys.path[:] = old_argv, old_path
        os.chdir(old_cwd)
        os.environ.pop("MANTA_GEN_TEST_DATA", None)
        for m in ("helperInclude", "helperGeneric"):
            sys.modules.pop(m, None)
    return buf.getvalue()


@pytest.mark.parametrize("name,min_checks", [("test_0010_io.py", 3), ("test_0011_inverted.py", 1), ("test_0020_shapes.py", 6),
                                             ("test_0040_interpol2d.py", 1), ("test_0041_interpol3d.py", 1),
                                             ("test_0030_gridop.py", 9), ("test_0100_psolve.py", 4), ("test_0150_advect.py", 10),
                                             ("test_1010_plume2d.py", 2), ("test_1070_flip2d.py", 2), ("test_1080_ldc.py", 1), ("test_2010_plume3d.py", 2),
                                             ("test_2011_plume3d_open.py", 2), ("test_2020_obstacle.py", 2),
                                             ("test_2070_falldropFlip.py", 1)])
def test_reference_regression_script(oracle_backend, tmp_path, name, min_checks):
    os.makedirs(tmp_path / "tests")
    os.makedirs(tmp_path / "testdata")
    out = run_script(name, str(tmp_path), gen=True)
    assert out.count("OK! Generated reference file") >= min_checks, out[-2000:]
    assert len(os.listdir(tmp_path / "testdata")) >= min_checks
    out = run_script(name, str(tmp_path), gen=False)
    assert "FAIL" not in out and "Error" not in out, out[-2000:]
    assert out.count("OK! Results for") >= min_checks, out[-2000:]
    shutil.rmtree(tmp_path / "testdata")
