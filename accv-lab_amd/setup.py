"""Packaging of the MI355X-native `accvlab.*` drop-in packages (namespace package `accvlab`, no __init__ at the top).

    pip install -e accv-lab_amd --no-build-isolation        # builds libaccv_hip.so + the host extensions in-tree first

The native parts are built by the Makefiles under csrc/ (hipcc --offload-arch=gfx950) and csrc_host/ (g++ against the
installed torch); this file only drives them and lists the package data.
"""
import os
import subprocess

from setuptools import find_namespace_packages, setup
from setuptools.command.build_py import build_py

HERE = os.path.dirname(os.path.abspath(__file__))


class BuildNative(build_py):
    def run(self):
        subprocess.run(["make", "-C", os.path.join(HERE, "csrc"), "-j", str(min(8, os.cpu_count() or 1))], check=True)
        subprocess.run(["make", "-C", os.path.join(HERE, "csrc_host")], check=True)
        super().run()


setup(
    name="accvlab-amd",
    version="0.1.0",
    description="MI355X-native (gfx950, hand-written HIP) drop-in for ACCV-Lab's draw_heatmap / batching_helpers / "
                "multi_tensor_copier / lane_helpers",
    packages=find_namespace_packages(where=HERE, include=["accvlab", "accvlab.*"]),
    package_dir={"": "."},
    package_data={"accvlab._amd_native": ["*.so"], "accvlab.multi_tensor_copier": ["*.so"],
                  "accvlab.batching_helpers": ["*.so"]},
    python_requires=">=3.9",
    install_requires=["torch", "numpy"],
    cmdclass={"build_py": BuildNative},
    zip_safe=False,
)
