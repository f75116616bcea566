#!/usr/bin/env python
"""source/test_proxe_s1.py of the reference: generation with the stage-1 model (see _gen_main.py)."""
from _gen_main import main_proxe

if __name__ == '__main__':
    main_proxe('s1')
