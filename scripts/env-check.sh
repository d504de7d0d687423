#!/bin/bash
python -m bigdl_amd.utils.env_check
