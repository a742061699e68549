"""dnnlib -- the two pieces of the reference's `dnnlib` that the custom-op hot path and
pickled model source rely on: `EasyDict` (reference dnnlib/util.py:40) and dotted-name
object lookup / construction (dnnlib/util.py:236-303). Everything else in the reference
package (URL cache, logger, file helpers) is outside the hot path and not provided."""

from .util import EasyDict, get_obj_by_name, call_func_by_name, construct_class_by_name, make_cache_dir_path, format_time

__all__ = ['EasyDict', 'get_obj_by_name', 'call_func_by_name', 'construct_class_by_name', 'make_cache_dir_path', 'format_time']
