"""Run configuration, mirroring the attribute bag of the reference (Sim_Config.py:6-51)."""
import datetime


class RL_Config(object):
    def __init__(self):
        self.Num_Feedback = 16        # Sim_Config.py:12
        self.Num_Episodes = 10000     # :13
        self.Num_Train_Steps = 20     # :14
        self.Batch_Size = 256         # :15
        self.Gamma = 0.2              # :16
        self.Num_Run_Episodes = 10    # :17
        self.Num_Test_Steps = 50      # :18
        self.Opt_Flag = True          # :19
        self.v2v_weight = 1           # :21
        self.v2i_weight = 1           # :23

    def set_train_value(self, num_feedback, gamma, batch_size, v2v_weight, v2i_weight):
        self.Num_Feedback, self.Gamma, self.Batch_Size = num_feedback, gamma, batch_size
        self.v2v_weight, self.v2i_weight = v2v_weight, v2i_weight

    def set_test_values(self, num_test_episodes, num_test_steps, opt_flag, v2v_weight, v2i_weight):
        self.Num_Run_Episodes, self.Num_Test_Steps, self.Opt_Flag = num_test_episodes, num_test_steps, opt_flag
        self.v2v_weight, self.v2i_weight = v2v_weight, v2i_weight

    def display(self):
        print(datetime.datetime.now().strftime('%Y/%m/%d %H:%M:%S'))
        print('Compression features %d, gamma %g, batch %d, episodes %d x %d train steps'
              % (self.Num_Feedback, self.Gamma, self.Batch_Size, self.Num_Episodes, self.Num_Train_Steps))
