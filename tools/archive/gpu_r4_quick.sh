#!/bin/bash
cd /root/repo
timeout -k 10 600 python -m pytest tests/test_physics_gpu.py -x -q -m gpu -k "destroy or refused or sub_ranges" 2>&1 | tail -15
