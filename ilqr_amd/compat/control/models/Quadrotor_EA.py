from ilqr_amd.models.Quadrotor_EA import QuadrotorEAModel  # noqa: F401
