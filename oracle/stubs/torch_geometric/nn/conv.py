from . import MessagePassing  # noqa: F401
