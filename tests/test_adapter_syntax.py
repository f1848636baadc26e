"""The real drop-in classes (csrc/hip_feature_matcher.h with -DMSF_WITH_SLAM_PIPELINE) compiled, syntax only, against
the reference's own plugin header /root/reference/slam_pipeline/include/FeatureMatcher.h:41-47 (and through it
FrameBase.h, KeyPointMap.h, types.h), with a 25-line stand-in for the OpenCV names those headers mention.  Build
container only: the GPU box has no /root/reference."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "slam_pipeline", "include", "FeatureMatcher.h")),
                    reason="the reference tree exists only in the build container")
def test_plugins_compile_against_the_reference_interface():
    cmd = ["g++", "-std=c++14", "-fsyntax-only", "-Wall", "-Werror=overloaded-virtual", "-DSLAM_PIPELINE_STATIC_DEFINE",
           "-I", os.path.join(ROOT, "tests", "cpp", "stubs"), "-I", REF, "-I", os.path.join(ROOT, "include"),
           "-I", os.path.join(ROOT, "mono_slam_framework_amd", "csrc"),
           os.path.join(ROOT, "tests", "cpp", "test_adapter_syntax.cpp")]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
