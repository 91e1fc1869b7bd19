"""Install `anemoi_transform_amd` (source directory `anemoi-transform_amd/`) and build libatx.so.

    pip install -e .        # needs hipcc (ROCm >= 7.0); builds for gfx950 only
"""

import os
import sys

from setuptools import setup
from setuptools.command.build_py import build_py

ROOT = os.path.dirname(os.path.abspath(__file__))


class BuildWithHip(build_py):
    def run(self):
        sys.path.insert(0, ROOT)
        import __graft_entry__ as graft

        graft.build()
        super().run()


setup(
    name="anemoi-transform-amd",
    version="0.4.2",
    description="MI355X-native (gfx950 HIP) filter hot path of ecmwf/anemoi-transform",
    packages=["anemoi_transform_amd", "anemoi_transform_amd.filters"],
    package_dir={"anemoi_transform_amd": "anemoi-transform_amd"},
    package_data={"anemoi_transform_amd": ["lib/libatx.so", "csrc/*"]},
    python_requires=">=3.10",
    install_requires=["numpy", "scipy", "torch"],
    cmdclass={"build_py": BuildWithHip},
)
