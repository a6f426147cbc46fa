#!/usr/bin/env python3
"""python run/train.py --config config/geopurify_synthetic_scannet.yaml [KEY VALUE ...]
Thin entry point; the driver lives in geopurify_amd/train_driver.py."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geopurify_amd.train_driver import main  # noqa: E402

if __name__ == "__main__":
    main()
