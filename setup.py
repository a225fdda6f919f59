"""Packaging for discorpy_amd: the HIP library is built in-tree by its Makefile (hipcc, gfx950) and shipped as
package data next to the Python host layer.  `python -c "import __graft_entry__ as g; g.build()"` does the same
build without installing anything."""
import os
import subprocess

from setuptools import find_packages, setup
from setuptools.command.build_py import build_py

ROOT = os.path.dirname(os.path.abspath(__file__))


class BuildWithHip(build_py):
    def run(self):
        subprocess.run(["make", "-C", os.path.join(ROOT, "discorpy_amd", "csrc"), "-j4", "ARCH=gfx950"], check=True)
        super().run()


setup(
    name="discorpy_amd",
    version="0.1.0",
    description="MI355X (gfx950) implementation of discorpy's backward unwarp path behind the same Python signatures",
    packages=find_packages(include=["discorpy_amd", "discorpy_amd.*"]),
    package_data={"discorpy_amd": ["lib/libdiscorpy_hip.so"]},
    python_requires=">=3.9",
    install_requires=["numpy"],
    cmdclass={"build_py": BuildWithHip},
)
