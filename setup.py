"""Packaging: builds the native libraries in-tree (g++ host library, hipcc gfx950 kernels) and installs the
``bigdl_amd`` package with the ``ipex_llm`` / ``bigdl`` alias packages and the console scripts.
Reference: python/llm/setup.py (extras :365-375, ``llm-convert`` console script :360-364)."""
import os
import sys

from setuptools import find_packages, setup
from setuptools.command.build_py import build_py

HERE = os.path.dirname(os.path.abspath(__file__))


class BuildNative(build_py):
    def run(self):
        sys.path.insert(0, HERE)
        from bigdl_amd.csrc.build import build_all
        build_all()
        super().run()


setup(
    name="bigdl-amd",
    version="0.1.0",
    description="MI355X-native low-bit LLM inference and fine-tuning library (IPEX-LLM compatible API)",
    packages=find_packages(include=["bigdl_amd", "bigdl_amd.*", "ipex_llm", "bigdl"]),
    package_data={"bigdl_amd": ["lib/*.so", "csrc/host/*", "csrc/hip/*"]},
    python_requires=">=3.10",
    install_requires=["torch", "transformers", "numpy", "safetensors", "tokenizers", "pyyaml"],
    extras_require={
        "serving": ["fastapi", "uvicorn", "pydantic", "prometheus_client"],
        "train": ["accelerate", "datasets"],
        "langchain": ["langchain-core"],
        "llama-index": ["llama-index-core"],
        "all": ["fastapi", "uvicorn", "pydantic", "accelerate", "datasets", "sentencepiece", "tabulate"],
    },
    entry_points={"console_scripts": [
        "llm-convert=bigdl_amd.convert_model:main",
        "llm-cli=bigdl_amd.cli.llm_cli:main",
        "llm-chat=bigdl_amd.cli.llm_cli:chat_main",
        "bigdl-amd-serve=bigdl_amd.serving.fastapi.api_server:main",
        "bigdl-amd-env-check=bigdl_amd.utils.env_check:main",
    ]},
    cmdclass={"build_py": BuildNative},
)
