// sql_stat.h -- one result row and the INSERT statement the apps print.
// The text format is kept byte-for-byte (reference: inc/sql_stat.h:28-79) so
// scripts/experiments/postprocessing/*.sh keep working: time in ms = ns/1e6,
// column order (time, correct, kernel, global, local, host, device, matrix,
// iteration, trial, statistic, experiment_id).
#pragma once
#include <chrono>
#include <sstream>
#include <string>
#include <vector>

enum Correctness { CORRECT, NOT_CHECKED, GENERIC_FAILURE, BAD_LENGTH, BAD_VALUES, GENERIC_BAD_VALUES, STATISTIC_VALUE };
enum TrialType { RAW_RESULT, MULTI_ITERATION_SUM, MEDIAN_RESULT };

class SqlStat {
public:
  SqlStat(std::chrono::nanoseconds time, Correctness correctness, unsigned int global, unsigned int local,
          TrialType trial_type, unsigned int trial = 0, unsigned int iteration = 0)
      : _time(time), _correctness(correctness), _global(global), _local(local), _trial_type(trial_type),
        _trial(trial), _iteration(iteration) {}

  std::chrono::nanoseconds getTime() const { return _time; }
  Correctness getCorrectness() const { return _correctness; }
  static bool compare(SqlStat a, SqlStat b) { return a.getTime() < b.getTime(); }
  static std::chrono::nanoseconds add(SqlStat a, SqlStat b) { return a.getTime() + b.getTime(); }

  std::string printStat(const std::string &kernel_name, const std::string &host_name,
                        const std::string &device_name, const std::string &matrix_name,
                        const std::string &experiment_id) const {
    std::ostringstream o;
    o << "(" << ((double)_time.count()) / 1000000.0 << ", \"" << correctnessName() << "\", \"" << kernel_name
      << "\", " << _global << ", " << _local << ", \"" << host_name << "\", \"" << device_name << "\", \""
      << matrix_name << "\"," << _iteration << "," << _trial << ",\"" << typeName() << "\", \"" << experiment_id
      << "\")";
    return o.str();
  }

  static std::string printHeader() {
    return "INSERT INTO table_name (time, correct, kernel, global, local, host, device, matrix, iteration, "
           "trial,statistic, experiment_id) VALUES ";
  }

  static std::string makeSqlCommand(const std::vector<SqlStat> &stats, const std::string &kernel_name,
                                    const std::string &host_name, const std::string &device_name,
                                    const std::string &matrix_name, const std::string &experiment_id) {
    std::string out = printHeader();
    for (std::size_t i = 0; i < stats.size(); i++) {
      if (i)
        out += ", ";
      out += stats[i].printStat(kernel_name, host_name, device_name, matrix_name, experiment_id);
    }
    return out + ";";
  }

private:
  const char *typeName() const {
    switch (_trial_type) {
    case RAW_RESULT: return "RAW_RESULT";
    case MULTI_ITERATION_SUM: return "MULTI_ITERATION_SUM";
    case MEDIAN_RESULT: return "MEDIAN_RESULT";
    }
    return "ERROR";
  }
  const char *correctnessName() const {
    switch (_correctness) {
    case CORRECT: return "correct";
    case NOT_CHECKED: return "notchecked";
    case GENERIC_FAILURE: return "genericfailure";
    case BAD_LENGTH: return "badlength";
    case BAD_VALUES:
    case GENERIC_BAD_VALUES: return "badvalues";
    case STATISTIC_VALUE: return "statisticvalue";
    }
    return "ERROR";
  }
  std::chrono::nanoseconds _time;
  Correctness _correctness;
  unsigned int _global, _local;
  TrialType _trial_type;
  unsigned int _trial, _iteration;
};
