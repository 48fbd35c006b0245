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


@pytest.mark.parametrize("pkg", ["nano_gicp", "quatro"])
def test_catkin_branch_puts_the_library_into_devel_space(tmp_path, pkg):
    """The catkin branch of the packages, configured against a stand-in catkin (tests/standins/cmake_catkin: devel prefix, destinations, a catkin_package()
    that resolves LIBRARIES under <devel>/lib like the generated <pkg>Config.cmake does): `catkin build` (devel space, the reference's README.md:75-78)
    finds libqn_engine.so because the prebuilt library is copied there at configure time."""
    if shutil.which("cmake") is None:
        pytest.skip("cmake not installed")
    lib = os.path.join(ROOT, "fast-lio-sam-qn_amd", "libqn_engine.so")
    if not os.path.exists(lib):
        pytest.skip("libqn_engine.so not built")
    b = tmp_path / "build"
    standin = os.path.join(ROOT, "tests", "standins", "cmake_catkin")
    subprocess.check_call(["cmake", "-S", os.path.join(ROOT, "packaging", pkg), "-B", str(b), "-DQN_ENGINE_ROOT=%s" % ROOT,
                           "-Dcatkin_DIR=%s" % standin, "-DPCL_DIR=%s" % standin, "-DEigen3_DIR=%s" % standin], stdout=subprocess.DEVNULL)
    assert (b / "devel" / "lib" / "libqn_engine.so").exists()
    args = (b / "devel" / "share" / pkg / "cmake" / ("%sConfig.cmake.args" % pkg)).read_text()
    assert "LIBRARIES=qn_engine" in args and "include" in args
