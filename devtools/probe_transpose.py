import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
from vdetlib_amd import _lib
cx = _lib.get_context(0)
print('wave_transpose verified:', cx.query(3), ' atomic_rank:', cx.query(0))
