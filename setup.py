"""setuptools entry point (the reference ships one next to its pyproject.toml,
reference: setup.py:1-12).  All metadata lives in pyproject.toml; this file adds the one thing
a declarative build cannot do: compiling the gfx950 kernel library and the host library
in-tree (``python -m minivllm_amd.build``) before the package data is collected.

    pip install -e .            # or: python setup.py build_ext --inplace

Both package roots are installed: ``minivllm_amd`` (the framework) and ``myvllm`` under
``src/`` (the reference's import paths, re-exporting the former)."""
from setuptools import find_packages, setup
from setuptools.command.build_py import build_py


class BuildWithNativeLibs(build_py):
    def run(self):
        try:
            from minivllm_amd.build import build_all
            build_all(force=False, verbose=True)
        except Exception as e:                   # no hipcc here: the Python parts still install
            print(f"[setup] native libraries were not built ({e}); run "
                  f"`python -m minivllm_amd.build` on a machine with ROCm")
        super().run()


setup(
    packages=find_packages(include=["minivllm_amd", "minivllm_amd.*"])
    + find_packages(where="src"),
    package_dir={"myvllm": "src/myvllm"},
    cmdclass={"build_py": BuildWithNativeLibs},
)
