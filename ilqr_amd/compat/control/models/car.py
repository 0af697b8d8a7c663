from ilqr_amd.models.car import CarModel  # noqa: F401
