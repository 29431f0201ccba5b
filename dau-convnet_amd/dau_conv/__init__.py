"""dau_conv: MI355X-native DAU convolution (drop-in for the reference's `dau_conv` package)."""
from .dau_conv import *  # noqa: F401,F403
from .dau_conv import (check_pending_offsets, constant_initializer, get_scope_layer, random_normal_initializer,  # noqa: F401
                       random_uniform_initializer, zeros_initializer)
from . import _capi  # noqa: F401
from ._capi import DAUConvError, FailedPreconditionError, InternalError, InvalidArgumentError  # noqa: F401
