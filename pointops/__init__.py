"""Drop-in name: ``import pointops`` resolves to the MI355X implementation.

The reference installs its CUDA extension under this package name (libs/pointops/setup.py:20-31,
``package_dir={"pointops": "functions"}``); putting this repository on PYTHONPATH (or installing it)
instead makes every ``import pointops`` / ``pointops.knn_query_and_group(...)`` in the reference's
``src/`` run on the HIP kernels with no source change.
"""
from pointcloudmatters_amd.pointops import *  # noqa: F401,F403
from pointcloudmatters_amd.pointops import __all__  # noqa: F401
