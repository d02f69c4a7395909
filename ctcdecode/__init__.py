"""Drop-in import name: ``from ctcdecode import CTCBeamDecoder, OnlineCTCBeamDecoder, DecoderState`` works unchanged
(reference: README.md:22-38, ctcdecode/__init__.py:6,143,253).  Everything lives in ``ctcdecode_amd``; this package only
re-exports it under the name the reference's users import."""
from ctcdecode_amd import CTCBeamDecoder, DecoderState, NativeError, OnlineCTCBeamDecoder  # noqa: F401

__all__ = ["CTCBeamDecoder", "OnlineCTCBeamDecoder", "DecoderState", "NativeError"]
