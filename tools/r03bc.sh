#!/bin/bash
timeout 900 python -m pytest tests/test_modules_gpu.py -q -x 2>&1 | grep -v "^$" | tail -45 | cut -c1-200
