"""``getattr(models, "SelfAttnRNNClassifier")`` target mirroring reference transfer/models.py."""
from .models import TransferSelfAttnRNNClassifier as SelfAttnRNNClassifier  # noqa: F401
