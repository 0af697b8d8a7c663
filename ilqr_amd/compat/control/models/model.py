from ilqr_amd.models.model import Model  # noqa: F401
