// Probe for mppi_amd::BufferedPlant (include/mppi_amd/plant.hpp): the history lists, getSmoothedBuffer's sampling and the
// per-cycle hook of runControlIteration — on a stub controller, so it runs without a device (mppi_lstm_lstm_initialize is host
// code).  The initialiser is the reference's known answer (tests/nn_helpers/lstm_lstm_helper_test.cu:161-180): LSTM(8, 60) +
// {68, 100, 20}, prediction hidden size 10, init_len 6, every parameter and every input 1 -> hidden = cell = 101.
#include <cmath>
#include <cstdio>
#include "mppi_amd/plant.hpp"

struct StubController
{
  int computes = 0, lstm_sets = 0, lstm_set_before_compute = 0;
  bool fresh_state = false;  // an initial state arrived since the last optimisation
  std::vector<float> hidden, cell;
  int getStateDim() const { return 2; }
  int getControlDim() const { return 1; }
  int getNumTimesteps() const { return 10; }
  float getDt() const { return 0.1f; }
  void updateImportanceSamplingControl(const std::vector<float>&, int) {}
  void slideControlSequence(int) {}
  void resetControls() {}
  void setLSTMInitialState(const std::vector<float>& h, const std::vector<float>& c)
  {
    hidden = h;
    cell = c;
    lstm_sets++;
    fresh_state = true;
  }
  void computeControl(const std::vector<float>&, int)
  {
    if (fresh_state)
      lstm_set_before_compute++;
    fresh_state = false;
    computes++;
  }
  mppi_stats getFreeEnergyStatistics() const { return mppi_stats(); }
  std::vector<float> getControlSeq() const { return std::vector<float>(10, 0.0f); }
  std::vector<float> getTargetStateSeq() const { return std::vector<float>(20, 0.0f); }
  std::vector<float> getTargetOutputSeq() const { return std::vector<float>(20, 0.0f); }
  std::vector<float> interpolateState(const std::vector<float>& traj, double) const { return { traj[0], traj[1] }; }
  std::vector<float> getCurrentControl(const std::vector<float>&, double, const std::vector<float>&, const std::vector<float>&,
                                       const std::vector<float>&) const { return { 0.0f }; }
};

struct Plant : mppi_amd::BufferedPlant<StubController>
{
  using mppi_amd::BufferedPlant<StubController>::BufferedPlant;
  void pubControl(const std::vector<float>&) override {}
  void pubNominalState(const std::vector<float>&) override {}
  void pubFreeEnergyStatistics(const mppi_stats&) override {}
  int checkStatus() override { return 0; }
  double getCurrentTime() override { return 0.0; }
  double getPoseTime() override { return 0.0; }
};

#define REQUIRE(c)                                        \
  do                                                      \
  {                                                       \
    if (!(c))                                             \
    {                                                     \
      fprintf(stderr, "FAILED line %d: %s\n", __LINE__, #c); \
      return 1;                                           \
    }                                                     \
  } while (0)

int main()
{
  auto ctl = std::make_shared<StubController>();
  Plant plant(ctl, 10, 1);
  std::atomic<bool> alive(true);
  // a ramp v(t) = 2 t sampled at irregular times: the smoothed buffer must be the ramp at its own regular times
  REQUIRE(!plant.checkRequiresBuffer());
  const double times[] = { 0.0, 0.03, 0.11, 0.2, 0.45, 0.46, 0.8, 0.95 };
  for (double t : times)
    plant.updateExtraValue("RAMP", (float)(2.0 * t), t);
  REQUIRE(plant.getSmoothedBuffer(0.95).empty());  // 0.95 s of history < tau = 1 s (buffer.hpp:216-224)
  plant.updateExtraValue("RAMP", 2.4f, 1.2);
  plant.updateExtraValue("RAMP", 0.0f, 0.5);  // older than the newest sample: dropped
  auto buf = plant.getSmoothedBuffer(1.2);
  REQUIRE(buf.count("RAMP") == 1 && buf["RAMP"].size() == 51);  // tau / dt + 1
  for (int i = 0; i <= 50; i++)
    REQUIRE(std::fabs(buf["RAMP"][i] - 2.0 * (1.2 - (50 - i) * 0.02)) < 1e-5);
  REQUIRE(std::fabs(plant.getInterpState(5.0)["RAMP"] - 2.4f) < 1e-6);  // clamped at the ends
  plant.cleanBuffers(2.9);  // horizon 2 s: everything before 0.9 s goes (the newest sample always stays)
  REQUIRE(std::fabs(plant.getInterpState(0.0)["RAMP"] - 1.9f) < 1e-6);
  plant.clearBuffers();

  // the LSTM hook: eight keys of ones, the all-ones initialiser
  mppi_amd::LSTMBufferInit li;
  li.init_input_dim = 8;
  li.init_hidden_dim = 60;
  li.init_output_layers = { 68, 100, 20 };
  li.hidden_dim = 10;
  li.init_len = 6;
  li.init_lstm_blob.assign(4 * 60 * 60 + 4 * 60 * 8 + 6 * 60, 1.0f);
  li.init_output_blob.assign(100 * 69 + 20 * 101, 1.0f);
  for (int i = 0; i < 8; i++)
  {
    li.keys.push_back("K" + std::to_string(i));
    li.scales.push_back(i < 2 ? 0.5f : 1.0f);
  }
  plant.setLSTMBufferInit(li);
  REQUIRE(plant.checkRequiresBuffer());
  // first cycle: no history yet -> the hook runs, finds nothing, the model keeps its state, the optimisation still happens
  plant.updateState({ 0.0f, 0.0f }, 0.0);
  plant.runControlIteration(&alive);
  REQUIRE(ctl->computes == 1 && ctl->lstm_sets == 0 && plant.numBufferUpdates() == 0);
  for (int k = 0; k <= 60; k++)
    for (int i = 0; i < 8; i++)
      plant.updateExtraValue(li.keys[i], i < 2 ? 2.0f : 1.0f, 0.02 * k);  // scaled by 0.5 -> ones
  plant.updateState({ 0.0f, 0.0f }, 1.2);
  plant.runControlIteration(&alive);
  REQUIRE(ctl->computes == 2 && ctl->lstm_sets == 1 && ctl->lstm_set_before_compute == 1 && plant.numBufferUpdates() == 1);
  REQUIRE(ctl->hidden.size() == 10 && ctl->cell.size() == 10);
  for (int i = 0; i < 10; i++)
    REQUIRE(ctl->hidden[i] == 101.0f && ctl->cell[i] == 101.0f);
  // a missing key leaves the model alone
  mppi_amd::LSTMBufferInit bad = li;
  bad.keys[3] = "NOT_THERE";
  plant.setLSTMBufferInit(bad);
  plant.updateState({ 0.0f, 0.0f }, 1.3);
  plant.runControlIteration(&alive);
  REQUIRE(ctl->computes == 3 && ctl->lstm_sets == 1);
  printf("BUFFERED PLANT OK\n");
  return 0;
}
