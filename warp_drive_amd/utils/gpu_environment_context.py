"""Glue between EnvWrapper and an environment class
(reference warp_drive/utils/gpu_environment_context.py:5-45).

The attribute names keep the reference's `cuda_` prefix so environment classes
written for WarpDrive run unchanged on the HIP backend."""
import logging

from warp_drive_amd.utils.data_feed import DataFeed


class CUDAEnvironmentContext:
    def __init__(self):
        self.cuda_data_manager = None
        self.cuda_function_manager = None
        self.cuda_step = None
        self.cuda_step_function_feed = None

    def initialize_step_function_context(self, cuda_data_manager, cuda_function_manager,
                                         cuda_step_function_feed, step_function_name):
        try:
            self.cuda_data_manager = cuda_data_manager
            self.cuda_function_manager = cuda_function_manager
            name = self.resolve_step_function_name(step_function_name)
            self.cuda_function_manager.initialize_functions([name])
            self.cuda_step = self.cuda_function_manager.get_function(name)
            self.cuda_step_function_feed = cuda_step_function_feed
            return True
        except Exception as err:  # same contract as the reference: report and return False
            logging.error(err)
            return False

    def resolve_step_function_name(self, default_name):
        """Hook: an env may pick a specialised kernel (e.g. a register-resident top-K)."""
        return default_name

    # data an env wants resident on the device; overridden by env classes
    def get_data_dictionary(self):
        return DataFeed()

    def get_tensor_dictionary(self):
        return DataFeed()

    def get_reset_pool_dictionary(self):
        return DataFeed()


HIPEnvironmentContext = CUDAEnvironmentContext
