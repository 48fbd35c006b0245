"""the replacement catkin packages `nano_gicp` / `quatro` (packaging/) configure and install with plain CMake (no
catkin in this image) and lay the headers out the way fast_lio_sam_qn/include/loop_closure.h:16-19 includes them."""
import os
import shutil
import subprocess
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("pkg,headers", [("nano_gicp", ["nano_gicp/nano_gicp.hpp", "nano_gicp/point_type_nano_gicp.hpp"]), ("quatro", ["quatro/quatro_module.h"])])
def test_package_configures_and_installs(tmp_path, pkg, headers):
    if shutil.which("cmake") is None:
        pytest.skip("cmake not installed")
    b, prefix = tmp_path / "build", tmp_path / "prefix"
    subprocess.check_call(["cmake", "-S", os.path.join(ROOT, "packaging", pkg), "-B", str(b), "-DCMAKE_INSTALL_PREFIX=%s" % prefix, "-DQN_ENGINE_ROOT=%s" % ROOT],
                          stdout=subprocess.DEVNULL)
    subprocess.check_call(["cmake", "--build", str(b)], stdout=subprocess.DEVNULL)
    subprocess.check_call(["cmake", "--install", str(b)], stdout=subprocess.DEVNULL)
    for h in headers + ["qn_engine.h"]:
        assert (prefix / "include" / h).exists(), h
    xml = open(os.path.join(ROOT, "packaging", pkg, "package.xml")).read()
    assert "<name>%s</name>" % pkg in xml and "<buildtool_depend>catkin</buildtool_depend>" in xml
