/**
 * lstm_lstm_helper.hpp — HOST side of the reference's LSTMLSTMHelper (include/mppi/utils/nn_helpers/lstm_lstm_helper.cuh,
 * lstm_lstm_helper.cu:4-12 constructor, :50-76 initializeLSTM): an "initialiser" LSTM that reads the last `init_len`
 * samples of a history buffer and whose output network emits the initial [hidden | cell] state of the prediction LSTM
 * evaluated inside the rollouts (utils/nn_helpers/lstm_helper.hpp on the device).
 *
 * The reference runs this on the CPU, once per control cycle, on the state-estimator side (Dynamics::updateFromBuffer ->
 * initializeLSTM -> copyHiddenCellToDevice); so does this class — plain C++, no device code, libm activations as the
 * reference's host flavour (utils/activation_functions.cuh host branches: tanhf, 1 / (1 + expf(-x))).  Its result goes to
 * the engine as the tail of the "lstm_weights" blob (mppi_set_model_blob), which is the analogue of
 * copyHiddenCellToDevice().
 *
 * Blob layouts are those of the device helpers: LSTM [W_im W_fm W_om W_cm | W_ii W_fi W_oi W_ci | b_i b_f b_o b_c | h0 | c0]
 * (gate order i, f, o, c; lstm_helper.cu:71-88), output network [W1 (out x in, row-major) | b1 | W2 | b2 ...].
 */
#ifndef MPPI_AMD_LSTM_LSTM_HELPER_HPP_
#define MPPI_AMD_LSTM_LSTM_HELPER_HPP_

#include <cmath>
#include <stdexcept>
#include <vector>

namespace mppi
{
/** host LSTM + output network with its own hidden / cell state (reference: LSTMHelper's CPU methods, lstm_helper.cu:266-323) */
class HostLSTM
{
public:
  int I = 0, H = 0;
  std::vector<int> out_layers;       ///< {H + I, ..., output dim}
  std::vector<float> lstm, out_net;  ///< parameter blobs (layouts above)
  std::vector<float> hidden, cell;

  HostLSTM() = default;
  HostLSTM(int input_dim, int hidden_dim, const std::vector<int>& output_layers)
    : I(input_dim), H(hidden_dim), out_layers(output_layers)
  {
    if (input_dim <= 0 || hidden_dim <= 0 || output_layers.size() < 2 || output_layers[0] != input_dim + hidden_dim)
      throw std::invalid_argument("HostLSTM: the output network's first layer must be hidden + input wide");
    lstm.assign(getNumParams(), 0.0f);
    out_net.assign(getOutputNumParams(), 0.0f);
    hidden.assign(H, 0.0f);
    cell.assign(H, 0.0f);
  }
  int getInputDim() const
  {
    return I;
  }
  int getHiddenDim() const
  {
    return H;
  }
  int getOutputDim() const
  {
    return out_layers.back();
  }
  /** LSTM blob size including h0, c0 */
  size_t getNumParams() const
  {
    return (size_t)4 * H * H + (size_t)4 * H * I + 6 * (size_t)H;
  }
  size_t getOutputNumParams() const
  {
    size_t n = 0;
    for (size_t l = 0; l + 1 < out_layers.size(); l++)
      n += (size_t)out_layers[l + 1] * out_layers[l] + out_layers[l + 1];
    return n;
  }
  /** reference: LSTMHelper::setAllValues(float) — every parameter, initial state included */
  void setAllValues(float v)
  {
    lstm.assign(lstm.size(), v);
    out_net.assign(out_net.size(), v);
  }
  void setAllValues(const float* lstm_blob, const float* out_blob)
  {
    lstm.assign(lstm_blob, lstm_blob + getNumParams());
    out_net.assign(out_blob, out_blob + getOutputNumParams());
  }
  /** reference: lstm_helper.cu:466-473 */
  void resetHiddenCellCPU()
  {
    const float* h0 = lstm.data() + (size_t)4 * H * H + (size_t)4 * H * I + 4 * (size_t)H;
    for (int i = 0; i < H; i++)
    {
      hidden[i] = h0[i];
      cell[i] = h0[H + i];
    }
  }
  /** reference: lstm_helper.cu:266-306 — the recurrent update alone */
  void forward(const float* x)
  {
    const float* W_m[4] = { lstm.data(), lstm.data() + H * H, lstm.data() + 2 * H * H, lstm.data() + 3 * H * H };
    const float* W_i[4];
    for (int g = 0; g < 4; g++)
      W_i[g] = lstm.data() + 4 * H * H + g * H * I;
    const float* b = lstm.data() + 4 * H * H + 4 * H * I;
    std::vector<float> h_next(H), c_next(H);
    for (int i = 0; i < H; i++)
    {
      float g[4];
      for (int k = 0; k < 4; k++)
      {
        float hh = 0.0f, xx = 0.0f;
        for (int j = 0; j < H; j++)
          hh += W_m[k][i * H + j] * hidden[j];
        for (int j = 0; j < I; j++)
          xx += W_i[k][i * I + j] * x[j];
        g[k] = hh + xx + b[k * H + i];
      }
      const float g_i = sigmoid(g[0]), g_f = sigmoid(g[1]), g_o = sigmoid(g[2]), g_c = std::tanh(g[3]);
      c_next[i] = g_i * g_c + g_f * cell[i];
      h_next[i] = g_o * std::tanh(c_next[i]);
    }
    hidden = h_next;
    cell = c_next;
  }
  /** reference: lstm_helper.cu:308-323 — update, then the output network on [h ; x] (tanh hidden layers, linear output) */
  void forward(const float* x, float* output)
  {
    forward(x);
    std::vector<float> act(hidden.begin(), hidden.end());
    act.insert(act.end(), x, x + I);
    size_t off = 0;
    for (size_t l = 0; l + 1 < out_layers.size(); l++)
    {
      const int n_in = out_layers[l], n_out = out_layers[l + 1];
      const float* W = out_net.data() + off;
      const float* bias = W + (size_t)n_out * n_in;
      std::vector<float> next(n_out);
      for (int j = 0; j < n_out; j++)
      {
        float acc = 0.0f;
        for (int k = 0; k < n_in; k++)
          acc += W[(size_t)j * n_in + k] * act[k];
        acc += bias[j];
        next[j] = (l + 2 < out_layers.size()) ? std::tanh(acc) : acc;
      }
      act.swap(next);
      off += (size_t)n_out * n_in + n_out;
    }
    for (int j = 0; j < getOutputDim(); j++)
      output[j] = act[j];
  }

private:
  static float sigmoid(float x)
  {
    return 1.0f / (1.0f + std::exp(-x));
  }
};

class LSTMLSTMHelper
{
public:
  /** reference: lstm_lstm_helper.cu:4-12 */
  LSTMLSTMHelper(int init_input_dim, int init_hidden_dim, const std::vector<int>& init_output_layers, int input_dim,
                 int hidden_dim, const std::vector<int>& output_layers, int init_len)
    : init_model_(init_input_dim, init_hidden_dim, init_output_layers)
    , input_dim_(input_dim)
    , hidden_dim_(hidden_dim)
    , output_layers_(output_layers)
    , init_len_(init_len)
  {
    if (init_model_.getOutputDim() != 2 * hidden_dim)
      throw std::invalid_argument("LSTMLSTMHelper: the initialiser must emit 2 x hidden values");
    if (init_len < 1)
      throw std::invalid_argument("LSTMLSTMHelper: init_len must be positive");
  }
  HostLSTM* getInitModel()
  {
    return &init_model_;
  }
  int getInitLen() const
  {
    return init_len_;
  }
  int getHiddenDim() const
  {
    return hidden_dim_;
  }
  /**
   * reference: lstm_lstm_helper.cu:50-76.  buffer: [init_input_dim][cols] with the rows contiguous per column, i.e. sample
   * t at buffer[t * init_input_dim ...] (Eigen's column-major MatrixXf with one column per time step); cols >= init_len.
   * The initialiser starts from its own (h0, c0), consumes the last init_len samples and its output after the last one
   * is [hidden | cell] of the prediction LSTM.
   */
  void initializeLSTM(const float* buffer, int cols, float* hidden_out, float* cell_out)
  {
    if (cols < init_len_)
      throw std::invalid_argument("LSTMLSTMHelper::initializeLSTM: buffer shorter than init_len");
    const int n_in = init_model_.getInputDim();
    init_model_.resetHiddenCellCPU();
    int t = cols - init_len_;
    for (; t < cols - 1; t++)
      init_model_.forward(buffer + (size_t)t * n_in);
    std::vector<float> output(init_model_.getOutputDim());
    init_model_.forward(buffer + (size_t)t * n_in, output.data());
    for (int i = 0; i < hidden_dim_; i++)
    {
      hidden_out[i] = output[i];
      cell_out[i] = output[hidden_dim_ + i];
    }
  }

private:
  HostLSTM init_model_;
  int input_dim_, hidden_dim_;
  std::vector<int> output_layers_;
  int init_len_;
};
}  // namespace mppi

#endif
