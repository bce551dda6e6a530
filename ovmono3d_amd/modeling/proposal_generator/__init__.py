from .rpn import RPNWithIgnore, NativeProposals  # noqa: F401
